// lyra_hip_components.h -- HIP-backed implementations of the reference's plugin interfaces and the factory
// functions that replace lyra/lyra_components.cc:42-55.  Each object is ONE stream of the shared GPU context.  Objects
// are used the way the reference's are (one hop per call, one thread per codec object); calls of the same kind that
// are waiting at the same time -- many codec objects on many threads -- are combined into ONE batched call of the C ABI
// (lyra_hip_components.cc "Call combining"), with results bit-identical to separate calls.  A caller that already holds
// a batch of streams uses BatchLyraEncoder / BatchLyraDecoder or the C ABI directly.
#ifndef LYRA_AMD_HOST_LYRA_HIP_COMPONENTS_H_
#define LYRA_AMD_HOST_LYRA_HIP_COMPONENTS_H_
#include <memory>

#include "include/ghc/filesystem.hpp"
#include "plugin_interfaces.h"

namespace chromemedia {
namespace codec {

// Same names / argument meaning as lyra/lyra_components.h:32-39.  `model_path` is a directory holding
// lyra_v1.lyrapack (tools/pack_weights.py output of the reference's model_coeffs).  nullptr on failure.
std::unique_ptr<VectorQuantizerInterface> CreateQuantizer(const ghc::filesystem::path& model_path);
std::unique_ptr<GenerativeModelInterface> CreateGenerativeModel(int num_output_features,
                                                                const ghc::filesystem::path& model_path);
std::unique_ptr<FeatureExtractorInterface> CreateFeatureExtractor(const ghc::filesystem::path& model_path);
// The NoiseEstimator front end (lyra/noise_estimator.cc:157-160): 16 kHz / hop 320 / window 640 / 160 mel bins.
std::unique_ptr<FeatureExtractorInterface> CreateLogMelExtractor(const ghc::filesystem::path& model_path);

// How the plugin calls of this process were served so far: calls made by plugin objects, device calls they became, and
// the largest number of streams one device call carried.
struct HipCallStats { long calls = 0, device_calls = 0, largest_batch = 0, gather_us = 0, exec_us = 0, gather_timeouts = 0; };
HipCallStats GetHipCallStats();
// GPU contexts the plugin objects of this process hold right now: 0 (no object alive), 1 (only extractor-side or only
// decoder-side calls so far) or 2.  Each holds the weights, the state of max_streams streams and its staging buffers.
int GetHipContextCount();

// Process-wide settings of the shared context (call before the first Create*).
void SetHipDevice(int device);
void SetMaxStreams(int max_streams);  // how many plugin objects may be alive at once (default 1024)

}  // namespace codec
}  // namespace chromemedia
#endif
