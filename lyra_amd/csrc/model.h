// model.h -- host side: weight container reader and the device-resident, MFMA-fragment-packed model.
// Replaces TfLiteModelWrapper (lyra/tflite_model_wrapper.cc:36-121): instead of building an interpreter
// over the flatbuffers, the coefficients are re-laid once at context creation for the hand-written kernels.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <string>
#include <vector>

#include "kernels.h"
#include "pack_format.h"

namespace lyra {

class Pack {
 public:
  bool open(const std::string& path, std::string* err);
  bool adopt(std::vector<uint8_t>&& image, std::string* err);   // an in-memory LYRAPK01 image (tflite_pack.h)
  const PackEntry* find(const std::string& name) const;
  // Payload of `name` iff it has exactly this dtype (0 f32, 1 i8, 2 i32) and shape; otherwise nullptr and the
  // first such failure is remembered (ok() / missing()).  The kernels are specialised to fixed layer shapes, so a
  // container with different ones is rejected here rather than over-read on the device.
  const void* get(const std::string& name, uint32_t dtype, std::initializer_list<uint32_t> shape) const;
  const float* f32(const std::string& name, std::initializer_list<uint32_t> shape) const {
    return static_cast<const float*>(get(name, 0, shape));
  }
  const int8_t* i8(const std::string& name, std::initializer_list<uint32_t> shape) const {
    return static_cast<const int8_t*>(get(name, 1, shape));
  }
  const int32_t* i32(const std::string& name, std::initializer_list<uint32_t> shape) const {
    return static_cast<const int32_t*>(get(name, 2, shape));
  }
  bool ok() const { return missing_.empty(); }
  const std::string& missing() const { return missing_; }

 private:
  std::vector<uint8_t> blob_;
  uint32_t n_ = 0;                 // validated entry count
  mutable std::string missing_;
};

struct Model {
  EncS0P enc0; EncS1P enc1; EncS2P enc2;
  DecS0P dec0; DecS1P dec1; DecS2P dec2;
  const float* cb = nullptr;   // [46][16][64]
  const float* cbt = nullptr;  // [46][64][16]
  const float* cbn = nullptr;  // [46][16] |c|^2 (rounded from double), then [46] 2^-14 max |c|^2 of the stage, rounded up: rvq_encode's screen
  MelP mel;            // = mel_rate[1]
  MelP mel_rate[4];    // log-mel tables of an extractor created for 8 / 16 / 32 / 48 kHz (model.hip)
  ResetP reset;
  // device-resident copies of the parameter blocks above (what the kernels actually read)
  EncS0P* d_enc0 = nullptr; EncS1P* d_enc1 = nullptr; EncS2P* d_enc2 = nullptr;
  DecS0P* d_dec0 = nullptr; DecS1P* d_dec1 = nullptr; DecS2P* d_dec2 = nullptr;
  MelP* d_mel = nullptr; ResetP* d_reset = nullptr;
  MelP* d_mel_rate[4] = {nullptr, nullptr, nullptr, nullptr};
  uint8_t* d_params = nullptr;
  uint8_t* d_arena = nullptr;
  size_t arena_bytes = 0;
};

// Builds the packed model on the current HIP device.  Returns false and sets *err on failure.
bool build_model(const Pack& pk, int requant_mode, Model* out, std::string* err);
void free_model(Model* m);

}  // namespace lyra
