// ref_dropin.cc -- TEST INFRASTRUCTURE: the LITERAL drop-in, built and run.
//
// oracle/Makefile target _ref/liblyra_ref_hip.so compiles, from /root/reference where they lie,
//
//   lyra/lyra_encoder.cc   (LyraEncoder::Create :43-96, Encode :113-156)
//   lyra/lyra_decoder.cc   (LyraDecoder::Create :97-155, SetEncodedPacket / DecodeSamples :172-373)
//   lyra/lyra_benchmark_lib.cc (lyra_benchmark :85-293, with -DBENCHMARK: its own timing table)
//   lyra/noise_estimator.cc, buffered_resampler.cc, lyra_config.cc, dsp_utils.cc,
//   lyra/cli_example/{encoder,decoder}_main_lib.cc (EncodeFile / DecodeFile), the two packet-loss models
//
// together with lyra_amd/host/lyra_hip_components.cc -- the product's plugin classes, here compiled against the
// reference's OWN interface headers (plugin_interfaces.h defers to them when they are on the include path) -- which
// supplies CreateFeatureExtractor / CreateQuantizer / CreateGenerativeModel in place of lyra_components.cc:42-55.
// Unlike oracle/ref_glue.cc nothing is injected through the *Peer friends: the reference's public Create() functions run
// (asset probe of lyra_config.h:117-168 included) and call the HIP factories themselves.  This file only adds
//   * the two factories a maintainer keeps as they are (lyra_components.cc:57-65; that file itself cannot be compiled
//     here because it includes the three TFLite-backed class headers), and
//   * a C API for the Python tests (tests/test_gpu_dropin.py).
// The classes that sit on un-vendored audio_dsp (log-mel front end of the NoiseEstimator, Resampler,
// ComfortNoiseGenerator) are the oracle-backed shadows of ref_shims/, exactly as in liblyra_ref.so: the hot path --
// the three networks -- is what is swapped, and every sample of it comes from liblyra_hip.so.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "absl/types/span.h"
#include "lyra/cli_example/decoder_main_lib.h"
#include "lyra/cli_example/encoder_main_lib.h"
#include "lyra/lyra_benchmark_lib.h"
#include "lyra/lyra_components.h"
#include "lyra/lyra_decoder.h"
#include "lyra/lyra_encoder.h"
#include "lyra/packet.h"
#include "lyra/zero_feature_estimator.h"
#include "lyra_hip_components.h"
#include "ref_oracle_api.h"

static const lo_model* g_model = nullptr;
static uint64_t g_cng_seed = 0;
const lo_model* ref_model() { return g_model; }          // read by the ref_shims/ shadows
uint64_t ref_next_cng_seed() { return g_cng_seed; }

namespace chromemedia {
namespace codec {
// lyra_components.cc:57-65, unchanged in a HIP-backed tree
std::unique_ptr<PacketInterface> CreatePacket(int num_header_bits, int num_quantized_bits) {
  return Packet<184>::Create(num_header_bits, num_quantized_bits);
}
std::unique_ptr<FeatureEstimatorInterface> CreateFeatureEstimator(int num_features) {
  return std::make_unique<ZeroFeatureEstimator>(num_features);
}
}  // namespace codec
}  // namespace chromemedia

using namespace chromemedia::codec;

extern "C" {

void dropin_set_oracle_model(const void* oracle_model) { g_model = (const lo_model*)oracle_model; }
void dropin_set_cng_seed(uint64_t seed) { g_cng_seed = seed; }
void dropin_set_max_streams(int n) { SetMaxStreams(n); }

// LyraEncoder::Create / LyraDecoder::Create as the reference's callers use them (lyra_encoder.h:61-63, lyra_decoder.h:54-56)
void* dropin_encoder_new(int sample_rate_hz, int num_channels, int bitrate, int enable_dtx, const char* model_path) {
  return LyraEncoder::Create(sample_rate_hz, num_channels, bitrate, enable_dtx != 0, model_path).release();
}
void dropin_encoder_free(void* e) { delete (LyraEncoder*)e; }
int dropin_encoder_encode(void* e, const int16_t* audio, int n, uint8_t* out, int cap) {   // -> packet bytes, -1 = nullopt
  auto r = ((LyraEncoder*)e)->Encode(absl::MakeConstSpan(audio, (size_t)n));
  if (!r.has_value() || (int)r->size() > cap) return -1;
  std::memcpy(out, r->data(), r->size());
  return (int)r->size();
}
int dropin_encoder_set_bitrate(void* e, int bitrate) { return ((LyraEncoder*)e)->set_bitrate(bitrate) ? 1 : 0; }

void* dropin_decoder_new(int sample_rate_hz, int num_channels, const char* model_path) {
  return LyraDecoder::Create(sample_rate_hz, num_channels, model_path).release();
}
void dropin_decoder_free(void* d) { delete (LyraDecoder*)d; }
int dropin_decoder_set_packet(void* d, const uint8_t* bytes, int n) {
  return ((LyraDecoder*)d)->SetEncodedPacket(absl::MakeConstSpan(bytes, (size_t)n)) ? 1 : 0;
}
int dropin_decoder_decode(void* d, int num_samples, int16_t* out) {   // -> samples written, -1 = std::nullopt
  auto r = ((LyraDecoder*)d)->DecodeSamples(num_samples);
  if (!r.has_value()) return -1;
  std::memcpy(out, r->data(), r->size() * 2);
  return (int)r->size();
}
int dropin_decoder_is_comfort_noise(void* d) { return ((LyraDecoder*)d)->is_comfort_noise() ? 1 : 0; }

// the reference's own benchmark loop (lyra_benchmark_lib.cc:199-293); its table goes to the log (stderr) and, as the
// reference does on desktop, to /tmp/benchmarks/*.csv
int dropin_lyra_benchmark(int num_cond_vectors, const char* model_path, int feature_extraction, int quantizer,
                          int generative_model) {
  return lyra_benchmark(num_cond_vectors, model_path, feature_extraction != 0, quantizer != 0, generative_model != 0);
}

// cli_example/encoder_main_lib.cc:99-140 / decoder_main_lib.cc:142-222: what encoder_main / decoder_main call
int dropin_encode_file(const char* wav_path, const char* out_path, int bitrate, int enable_preprocessing, int enable_dtx,
                       const char* model_path) {
  return EncodeFile(wav_path, out_path, bitrate, enable_preprocessing != 0, enable_dtx != 0, model_path) ? 1 : 0;
}
int dropin_decode_file(const char* encoded_path, const char* out_path, int sample_rate_hz, int bitrate,
                       const char* model_path) {
  const PacketLossPattern none({}, {});
  return DecodeFile(encoded_path, out_path, sample_rate_hz, bitrate, false, 0.f, 1.f, none, model_path) ? 1 : 0;
}

// how the plugin calls were served (lyra_hip_components.h)
void dropin_call_stats(long* calls, long* device_calls) {
  const HipCallStats s = GetHipCallStats();
  *calls = s.calls;
  *device_calls = s.device_calls;
}

}  // extern "C"
