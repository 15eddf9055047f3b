// plugin_interfaces.h -- the three plugin ABCs of the reference's hot path, same names / signatures / semantics,
// so that code written against google/lyra's lyra_components compiles against the HIP-backed implementations:
//   FeatureExtractorInterface   lyra/feature_extractor_interface.h:32-39
//   VectorQuantizerInterface    lyra/vector_quantizer_interface.h:28-41
//   GenerativeModelInterface + GenerativeModel FIFO base   lyra/generative_model_interface.h:32-134
// (Interfaces are API, not implementation: they have to be spelled the same way to be a drop-in.)
//
// Inside a google/lyra checkout the reference's own headers are used instead (the #if below); this file is the
// stand-alone spelling for builds without it.  The three abstract classes and the GenerativeModel FIFO base restate
// declarations of google/lyra, Copyright 2021 Google LLC, licensed under the Apache License, Version 2.0
// (http://www.apache.org/licenses/LICENSE-2.0); used here under that licence, "AS IS", without warranties or conditions
// of any kind.
#ifndef LYRA_AMD_HOST_PLUGIN_INTERFACES_H_
#define LYRA_AMD_HOST_PLUGIN_INTERFACES_H_
#if defined(__has_include) && __has_include("lyra/generative_model_interface.h") && \
    __has_include("lyra/feature_extractor_interface.h") && __has_include("lyra/vector_quantizer_interface.h")
#include "lyra/feature_extractor_interface.h"
#include "lyra/generative_model_interface.h"
#include "lyra/vector_quantizer_interface.h"
#else
#include <cstdint>
#include <optional>
#include <queue>
#include <string>
#include <vector>

#include "absl/types/span.h"
#include "glog/logging.h"

namespace chromemedia {
namespace codec {

class FeatureExtractorInterface {
 public:
  virtual ~FeatureExtractorInterface() {}
  // One hop of int16 audio -> feature vector; nullopt on failure.
  virtual std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) = 0;
};

class VectorQuantizerInterface {
 public:
  virtual ~VectorQuantizerInterface() {}
  // Bit string of '0'/'1', first quantizer in the most significant position.
  virtual std::optional<std::string> Quantize(const std::vector<float>& features, int num_bits) const = 0;
  virtual std::optional<std::vector<float>> DecodeToLossyFeatures(const std::string& quantized_features) const = 0;
};

class GenerativeModelInterface {
 public:
  virtual ~GenerativeModelInterface() {}
  virtual bool AddFeatures(const std::vector<float>& features) = 0;
  virtual std::optional<std::vector<int16_t>> GenerateSamples(int num_samples) = 0;
  virtual int num_samples_available() const = 0;
};

// FIFO / partial-hop state machine shared by generative models (generative_model_interface.h:45-134):
// features are queued; conditioning runs once when a hop starts; requests never straddle a hop.
class GenerativeModel : public GenerativeModelInterface {
 public:
  bool AddFeatures(const std::vector<float>& features) override final {
    if (static_cast<int>(features.size()) != num_features_) {
      LOG(ERROR) << "Expecting features to be of shape " << num_features_ << " but were of shape "
                 << features.size() << ".";
      return false;
    }
    queue_.push(features);
    return true;
  }
  std::optional<std::vector<int16_t>> GenerateSamples(int num_samples) override final {
    if (num_samples < 0) { LOG(ERROR) << "Number of samples must be positive."; return std::nullopt; }
    if (num_samples == 0) return std::vector<int16_t>();
    if (num_samples_available() == 0) {
      LOG(ERROR) << "Tried generating " << num_samples << " samples but only 0 are available.";
      return std::nullopt;
    }
    if (next_sample_in_hop_ == 0 && !RunConditioning(queue_.front())) return std::nullopt;
    const int remaining = num_samples_per_hop_ - next_sample_in_hop_;
    if (num_samples > remaining) {
      LOG(ERROR) << "Tried generating " << num_samples << " samples but only " << remaining
                 << " were available in current features.";
      return std::nullopt;
    }
    auto samples = RunModel(num_samples);
    if (samples.has_value()) {
      next_sample_in_hop_ += static_cast<int>(samples->size());
      if (next_sample_in_hop_ == num_samples_per_hop_) { next_sample_in_hop_ = 0; queue_.pop(); }
    }
    return samples;
  }
  int num_samples_available() const override final {
    return static_cast<int>(queue_.size()) * num_samples_per_hop_ - next_sample_in_hop_;
  }

 protected:
  GenerativeModel(int num_samples_per_hop, int num_features)
      : num_samples_per_hop_(num_samples_per_hop), num_features_(num_features), next_sample_in_hop_(0) {}
  virtual bool RunConditioning(const std::vector<float>& features) = 0;
  virtual std::optional<std::vector<int16_t>> RunModel(int num_samples) = 0;
  int next_sample_in_hop() const { return next_sample_in_hop_; }

 private:
  const int num_samples_per_hop_;
  const int num_features_;
  int next_sample_in_hop_;
  std::queue<std::vector<float>> queue_;
};

}  // namespace codec
}  // namespace chromemedia
#endif  // reference headers not on the include path
#endif
