// SHADOW of lyra/log_mel_spectrogram_extractor_impl.h (the reference's sits on un-vendored audio_dsp::Spectrogram /
// MelFilterbank): same public interface (log_mel_spectrogram_extractor_impl.h:33-62), Extract() computed by the CPU
// oracle's log-mel, which IS pinned to the reference's own 3x10 golden (tests/test_oracle_golden.py).
#ifndef REF_SHADOW_LOG_MEL_SPECTROGRAM_EXTRACTOR_IMPL_H_
#define REF_SHADOW_LOG_MEL_SPECTROGRAM_EXTRACTOR_IMPL_H_
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

#include "absl/types/span.h"
#include "lyra/feature_extractor_interface.h"
#include "ref_oracle_api.h"

namespace chromemedia {
namespace codec {

class LogMelSpectrogramExtractorImpl : public FeatureExtractorInterface {
 public:
  static std::unique_ptr<LogMelSpectrogramExtractorImpl> Create(int sample_rate_hz, int hop_length_samples,
                                                                int window_length_samples, int num_mel_bins) {
    // the oracle's log-mel is the 320 / 640 / 160 instance every caller on this path asks for; the sample rate selects the
    // mel filterbank (log_mel_spectrogram_extractor_impl.cc:81-87) -- the DTX encoder's NoiseEstimator passes its
    // EXTERNAL rate here (lyra_encoder.cc:82-85, noise_estimator.cc:104-106)
    if (hop_length_samples != 320 || window_length_samples != 640 || num_mel_bins != 160) return nullptr;
    if (sample_rate_hz != 8000 && sample_rate_hz != 16000 && sample_rate_hz != 32000 && sample_rate_hz != 48000) return nullptr;
    return std::unique_ptr<LogMelSpectrogramExtractorImpl>(new LogMelSpectrogramExtractorImpl(sample_rate_hz));
  }
  ~LogMelSpectrogramExtractorImpl() override { lo_stream_free(state_); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (audio.size() != 320) return std::nullopt;
    std::vector<float> mel(160);
    lo_logmel_rate(ref_model(), state_, audio.data(), mel.data(), sample_rate_hz_);
    return mel;
  }
  static float GetNormalizationFactor() { return 10.f; }
  static float GetSilenceValue() { return std::log(500.f) / 10.f; }

 private:
  explicit LogMelSpectrogramExtractorImpl(int sample_rate_hz) : state_(lo_stream_new()), sample_rate_hz_(sample_rate_hz) {}
  lo_stream* state_;
  int sample_rate_hz_;
};

}  // namespace codec
}  // namespace chromemedia
#endif
