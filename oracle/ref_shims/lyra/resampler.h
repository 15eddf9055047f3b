// SHADOW of lyra/resampler.h (the reference's wraps un-vendored audio_dsp::QResampler<float>): same public interface
// (resampler.h:30-52), Resample() computed by the CPU oracle's restatement of that polyphase resampler.
#ifndef REF_SHADOW_RESAMPLER_H_
#define REF_SHADOW_RESAMPLER_H_
#include <cstdint>
#include <memory>
#include <vector>

#include "absl/types/span.h"
#include "lyra/resampler_interface.h"
#include "ref_oracle_api.h"

namespace chromemedia {
namespace codec {

class Resampler : public ResamplerInterface {
 public:
  ~Resampler() override { lo_resampler_free(r_); }
  static std::unique_ptr<Resampler> Create(int input_sample_rate_hz, int target_sample_rate_hz) {
    lo_resampler* r = lo_resampler_new(input_sample_rate_hz, target_sample_rate_hz);
    if (!r) return nullptr;
    return std::unique_ptr<Resampler>(new Resampler(r, input_sample_rate_hz, target_sample_rate_hz));
  }
  std::vector<int16_t> Resample(absl::Span<const int16_t> audio) override {
    std::vector<int16_t> out(audio.size() * (size_t)out_ / (size_t)in_ + 8);
    const int n = lo_resample(r_, audio.data(), (int)audio.size(), out.data());
    out.resize(n < 0 ? 0 : n);
    return out;
  }
  void Reset() override { lo_resampler_reset(r_); }
  int input_sample_rate_hz() const override { return in_; }
  int target_sample_rate_hz() const override { return out_; }
  int samples_until_steady_state() const override { return 0; }

 private:
  Resampler(lo_resampler* r, int in, int out) : r_(r), in_(in), out_(out) {}
  lo_resampler* r_;
  const int in_, out_;
};

}  // namespace codec
}  // namespace chromemedia
#endif
