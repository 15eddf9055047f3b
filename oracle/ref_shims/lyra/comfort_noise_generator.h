// SHADOW of lyra/comfort_noise_generator.h (the reference's sits on un-vendored audio_dsp::MelFilterbank /
// InverseSpectrogram and draws phases from absl::BitGen): a GenerativeModel -- the reference's OWN FIFO base class,
// generative_model_interface.h:45-134 -- whose hop is the CPU oracle's comfort noise (counter-based phases, seeded).
#ifndef REF_SHADOW_COMFORT_NOISE_GENERATOR_H_
#define REF_SHADOW_COMFORT_NOISE_GENERATOR_H_
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

#include "lyra/generative_model_interface.h"
#include "ref_oracle_api.h"

namespace chromemedia {
namespace codec {

class ComfortNoiseGenerator : public GenerativeModel {
 public:
  static std::unique_ptr<ComfortNoiseGenerator> Create(int sample_rate_hz, int num_samples_per_hop,
                                                       int window_length_samples, int num_mel_bins) {
    if (sample_rate_hz != 16000 || num_samples_per_hop != 320 || window_length_samples != 640 || num_mel_bins != 160)
      return nullptr;
    return std::unique_ptr<ComfortNoiseGenerator>(new ComfortNoiseGenerator(ref_next_cng_seed()));
  }
  ~ComfortNoiseGenerator() override { lo_cng_free(cng_); }

 private:
  explicit ComfortNoiseGenerator(uint64_t seed) : GenerativeModel(320, 160), cng_(lo_cng_new(seed)), hop_(320) {}
  bool RunConditioning(const std::vector<float>& features) override {
    lo_cng_generate(ref_model(), cng_, features.data(), hop_.data());
    return true;
  }
  std::optional<std::vector<int16_t>> RunModel(int num_samples) override {
    return std::vector<int16_t>(hop_.begin() + next_sample_in_hop(), hop_.begin() + next_sample_in_hop() + num_samples);
  }
  lo_cng* cng_;
  std::vector<int16_t> hop_;
};

}  // namespace codec
}  // namespace chromemedia
#endif
