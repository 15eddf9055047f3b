// ref_glue.cc -- TEST INFRASTRUCTURE: drives the REFERENCE'S OWN host classes, compiled from /root/reference where they
// lie (oracle/Makefile, target _ref/liblyra_ref.so), through a small C API for the Python tests:
//
//   LyraDecoder  (lyra/lyra_decoder.cc:172-373)      packet FIFO, DecodeSamples(n), concealment, comfort noise, fades
//   LyraEncoder  (lyra/lyra_encoder.cc:113-156)      resample, DTX, extract, quantize, pack
//   NoiseEstimator (lyra/noise_estimator.cc:144-245) with its log-mel front end injected (ref_shims/ shadow)
//   BufferedResampler (lyra/buffered_resampler.cc)   leftover bookkeeping of the decoder's output resampler
//   Packet<> (lyra/packet.h:91-146), GenerativeModel FIFO (generative_model_interface.h:45-134), ZeroFeatureEstimator
//   EncodeFile / DecodeFile (cli_example/encoder_main_lib.cc, decoder_main_lib.cc) + the two packet-loss models
//   Int16ToUnit / UnitToInt16 / ClipToInt16 / LogSpectralDistance (lyra/dsp_utils.h:41-108, dsp_utils.cc:27-41)
//
// What cannot be compiled here -- TFLite + XNNPACK behind the three model wrappers, audio_dsp behind the resampler, the
// log-mel extractor and the comfort-noise generator -- is injected as components computed by the CPU oracle
// (oracle/lyra_oracle.c): the private constructors are reached through the *Peer friends the reference declares for its
// own tests (lyra_decoder.h:162, lyra_encoder.h:121).  So the CONTROL FLOW, the packet layout, the noise-estimator
// recurrence and the resampler buffering checked against this library are the reference's, statement for statement;
// the arithmetic of the networks stays the oracle's (parity of that part is pinned elsewhere).
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "absl/types/span.h"
#include "lyra/buffered_resampler.h"
#include "lyra/cli_example/decoder_main_lib.h"
#include "lyra/cli_example/encoder_main_lib.h"
#include "lyra/dsp_utils.h"
#include "lyra/comfort_noise_generator.h"
#include "lyra/feature_extractor_interface.h"
#include "lyra/generative_model_interface.h"
#include "lyra/lyra_components.h"
#include "lyra/lyra_config.h"
#include "lyra/lyra_decoder.h"
#include "lyra/lyra_encoder.h"
#include "lyra/noise_estimator.h"
#include "lyra/packet.h"
#include "lyra/resampler.h"
#include "lyra/vector_quantizer_interface.h"
#include "lyra/zero_feature_estimator.h"
#include "ref_oracle_api.h"

static const lo_model* g_model = nullptr;
static uint64_t g_cng_seed = 0;
const lo_model* ref_model() { return g_model; }
uint64_t ref_next_cng_seed() { return g_cng_seed; }

namespace chromemedia {
namespace codec {
namespace {

constexpr int kHop = 320, kFeat = 64, kBitsPerStage = 4, kMaxBits = 184;

// lyra_gan_model.cc:53-64 with the oracle's decoder in place of the TFLite interpreter
class OracleGan : public GenerativeModel {
 public:
  OracleGan() : GenerativeModel(kHop, kFeat), s_(lo_stream_new()), hop_(kHop) {}
  ~OracleGan() override { lo_stream_free(s_); }

 private:
  bool RunConditioning(const std::vector<float>& features) override {
    lo_decode_frame(g_model, s_, features.data(), hop_.data(), nullptr);
    return true;
  }
  std::optional<std::vector<int16_t>> RunModel(int num_samples) override {
    return std::vector<int16_t>(hop_.begin() + next_sample_in_hop(), hop_.begin() + next_sample_in_hop() + num_samples);
  }
  lo_stream* s_;
  std::vector<int16_t> hop_;
};

// soundstream_encoder.cc:53-64
class OracleExtractor : public FeatureExtractorInterface {
 public:
  OracleExtractor() : s_(lo_stream_new()) {}
  ~OracleExtractor() override { lo_stream_free(s_); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (audio.size() != kHop) return std::nullopt;
    std::vector<float> f(kFeat);
    lo_encode_frame(g_model, s_, audio.data(), f.data());
    return f;
  }

 private:
  lo_stream* s_;
};

// residual_vector_quantizer.cc:77-168: stage indices <-> a string of '0' / '1', first stage in the leading characters
class OracleQuantizer : public VectorQuantizerInterface {
 public:
  std::optional<std::string> Quantize(const std::vector<float>& features, int num_bits) const override {
    if (num_bits > kMaxBits || num_bits % kBitsPerStage != 0 || (int)features.size() != kFeat) return std::nullopt;
    const int stages = num_bits / kBitsPerStage;
    int32_t idx[46];
    lo_rvq_encode(g_model, features.data(), stages, idx);
    std::string bits((size_t)num_bits, '0');
    for (int i = 0; i < stages; ++i)
      for (int b = 0; b < kBitsPerStage; ++b)
        if ((idx[i] >> (kBitsPerStage - 1 - b)) & 1) bits[(size_t)(i * kBitsPerStage + b)] = '1';
    return bits;
  }
  std::optional<std::vector<float>> DecodeToLossyFeatures(const std::string& q) const override {
    const int num_bits = (int)q.size();
    if (num_bits > kMaxBits || num_bits % kBitsPerStage != 0) return std::nullopt;
    const int stages = num_bits / kBitsPerStage;
    int32_t idx[46];
    for (int i = 0; i < 46; ++i) idx[i] = -1;
    for (int i = 0; i < stages; ++i) {
      int v = 0;
      for (int b = 0; b < kBitsPerStage; ++b) v = (v << 1) | (q[(size_t)(i * kBitsPerStage + b)] == '1');
      idx[i] = v;
    }
    std::vector<float> f(kFeat);
    lo_rvq_decode(g_model, idx, f.data());
    return f;
  }
};

}  // namespace

// ---- the factories of lyra/lyra_components.h (lyra_components.cc:42-65 builds the TFLite-backed ones) ---------------
std::unique_ptr<VectorQuantizerInterface> CreateQuantizer(const ghc::filesystem::path&) {
  return std::make_unique<OracleQuantizer>();
}
std::unique_ptr<GenerativeModelInterface> CreateGenerativeModel(int, const ghc::filesystem::path&) {
  return std::make_unique<OracleGan>();
}
std::unique_ptr<FeatureExtractorInterface> CreateFeatureExtractor(const ghc::filesystem::path&) {
  return std::make_unique<OracleExtractor>();
}
std::unique_ptr<PacketInterface> CreatePacket(int num_header_bits, int num_quantized_bits) {
  return Packet<kMaxBits>::Create(num_header_bits, num_quantized_bits);
}
std::unique_ptr<FeatureEstimatorInterface> CreateFeatureEstimator(int num_features) {
  return std::make_unique<ZeroFeatureEstimator>(num_features);
}

// ---- the friends the reference declares for its tests: the way to the private constructors ---------------------------
class LyraDecoderPeer {
 public:
  // LyraDecoder::Create (lyra_decoder.cc:97-155) minus AreParamsSupported's probe for *.tflite files
  static std::unique_ptr<LyraDecoder> Make(int sample_rate_hz) {
    if (!IsSampleRateSupported(sample_rate_hz)) return nullptr;
    auto resampler = BufferedResampler::Create(kInternalSampleRateHz, sample_rate_hz);
    auto cng = ComfortNoiseGenerator::Create(kInternalSampleRateHz, GetNumSamplesPerHop(kInternalSampleRateHz),
                                             GetNumSamplesPerWindow(kInternalSampleRateHz), kNumMelBins);
    auto noise = NoiseEstimator::Create(kInternalSampleRateHz, GetNumSamplesPerHop(kInternalSampleRateHz),
                                        GetNumSamplesPerWindow(kInternalSampleRateHz), kNumMelBins);
    if (!resampler || !cng || !noise) return nullptr;
    return std::unique_ptr<LyraDecoder>(new LyraDecoder(CreateGenerativeModel(kNumFeatures, ""), std::move(cng),
                                                        CreateQuantizer(""), std::move(noise),
                                                        CreateFeatureEstimator(kNumFeatures), std::move(resampler),
                                                        sample_rate_hz, kNumChannels));
  }
};
class LyraEncoderPeer {
 public:
  // LyraEncoder::Create (lyra_encoder.cc:43-96) minus the file probe
  static std::unique_ptr<LyraEncoder> Make(int sample_rate_hz, int bitrate, bool enable_dtx) {
    if (!IsSampleRateSupported(sample_rate_hz)) return nullptr;
    const int num_quantized_bits = BitrateToNumQuantizedBits(bitrate);
    if (num_quantized_bits < 0) return nullptr;
    std::unique_ptr<Resampler> resampler;
    if (kInternalSampleRateHz != sample_rate_hz) resampler = Resampler::Create(sample_rate_hz, kInternalSampleRateHz);
    std::unique_ptr<NoiseEstimatorInterface> noise;
    if (enable_dtx)   // (the reference passes sample_rate_hz here; only the hop duration is derived from it)
      noise = NoiseEstimator::Create(sample_rate_hz, GetNumSamplesPerHop(kInternalSampleRateHz),
                                     GetNumSamplesPerWindow(kInternalSampleRateHz), kNumMelBins);
    return std::unique_ptr<LyraEncoder>(new LyraEncoder(std::move(resampler), CreateFeatureExtractor(""), std::move(noise),
                                                        CreateQuantizer(""), sample_rate_hz, kNumChannels,
                                                        num_quantized_bits, enable_dtx));
  }
};
class NoiseEstimatorPeer {
 public:
  static std::vector<float> bound(const NoiseEstimator& n) { return n.noise_bound_; }
};

}  // namespace codec
}  // namespace chromemedia

// ---- C API for ctypes ------------------------------------------------------------------------------------------------
using namespace chromemedia::codec;

extern "C" {

void ref_set_model(const void* oracle_model) { g_model = (const lo_model*)oracle_model; }
void ref_set_cng_seed(uint64_t seed) { g_cng_seed = seed; }   // taken by the next decoder's ComfortNoiseGenerator::Create

void* ref_decoder_new(int sample_rate_hz) { return LyraDecoderPeer::Make(sample_rate_hz).release(); }
void ref_decoder_free(void* d) { delete (LyraDecoder*)d; }
int ref_decoder_set_packet(void* d, const uint8_t* bytes, int n) {
  return ((LyraDecoder*)d)->SetEncodedPacket(absl::MakeConstSpan(bytes, (size_t)n)) ? 1 : 0;
}
int ref_decoder_decode(void* d, int num_samples, int16_t* out) {   // -> samples written, -1 = std::nullopt
  auto r = ((LyraDecoder*)d)->DecodeSamples(num_samples);
  if (!r.has_value()) return -1;
  std::memcpy(out, r->data(), r->size() * 2);
  return (int)r->size();
}
int ref_decoder_is_comfort_noise(void* d) { return ((LyraDecoder*)d)->is_comfort_noise() ? 1 : 0; }

void* ref_encoder_new(int sample_rate_hz, int bitrate, int enable_dtx) {
  return LyraEncoderPeer::Make(sample_rate_hz, bitrate, enable_dtx != 0).release();
}
void ref_encoder_free(void* e) { delete (LyraEncoder*)e; }
int ref_encoder_encode(void* e, const int16_t* audio, int n, uint8_t* out, int cap) {   // -> packet bytes, -1 = nullopt
  auto r = ((LyraEncoder*)e)->Encode(absl::MakeConstSpan(audio, (size_t)n));
  if (!r.has_value() || (int)r->size() > cap) return -1;
  std::memcpy(out, r->data(), r->size());
  return (int)r->size();
}
int ref_encoder_set_bitrate(void* e, int bitrate) { return ((LyraEncoder*)e)->set_bitrate(bitrate) ? 1 : 0; }

void* ref_noise_new(void) {
  return NoiseEstimator::Create(16000, 320, 640, 160).release();
}
// what a DTX LyraEncoder created at `sample_rate_hz` builds (lyra_encoder.cc:82-85): external rate, internal hop / window
void* ref_noise_new_rate(int sample_rate_hz) {
  return NoiseEstimator::Create(sample_rate_hz, 320, 640, 160).release();
}
void ref_noise_free(void* n) { delete (NoiseEstimator*)n; }
int ref_noise_receive(void* n, const int16_t* pcm, int count) {   // -> is_noise after the call, -1 on failure
  NoiseEstimator* ne = (NoiseEstimator*)n;
  if (!ne->ReceiveSamples(absl::MakeConstSpan(pcm, (size_t)count))) return -1;
  return ne->is_noise() ? 1 : 0;
}
void ref_noise_get(void* n, float* estimate, float* bound) {
  NoiseEstimator* ne = (NoiseEstimator*)n;
  const auto e = ne->noise_estimate();
  if (estimate) std::memcpy(estimate, e.data(), e.size() * 4);
  if (bound) { const auto b = NoiseEstimatorPeer::bound(*ne); std::memcpy(bound, b.data(), b.size() * 4); }
}

// Packet<>::PackQuantized / UnpackPacket (packet.h:91-146) on a string of '0' / '1'
int ref_packet_pack(const char* bits, uint8_t* out, int cap) {
  const std::string q(bits);
  auto p = CreatePacket(kNumHeaderBits, (int)q.size());
  if (!p) return -1;
  const auto bytes = p->PackQuantized(q);
  if ((int)bytes.size() > cap) return -1;
  std::memcpy(out, bytes.data(), bytes.size());
  return (int)bytes.size();
}
int ref_packet_unpack(const uint8_t* bytes, int n, char* bits_out, int cap) {
  const int nbits = PacketSizeToNumQuantizedBits(n);
  if (nbits < 0 || nbits + 1 > cap) return -1;
  auto p = CreatePacket(kNumHeaderBits, nbits);
  const auto q = p->UnpackPacket(absl::MakeConstSpan(bytes, (size_t)n));
  if (!q.has_value()) return -1;
  std::memcpy(bits_out, q->c_str(), q->size() + 1);
  return (int)q->size();
}
int ref_packet_size_to_bits(int packet_size) { return PacketSizeToNumQuantizedBits(packet_size); }
int ref_bitrate_to_bits(int bitrate) { return BitrateToNumQuantizedBits(bitrate); }
const char* ref_version(void) { return GetVersionString().c_str(); }

// dsp_utils.h:54-108 and dsp_utils.cc:27-41, as the model wrappers and the integration test use them
void ref_unit_to_int16(const float* in, long n, int16_t* out) { for (long i = 0; i < n; ++i) out[i] = UnitToInt16Scalar(in[i]); }
void ref_int16_to_unit(const int16_t* in, long n, float* out) { for (long i = 0; i < n; ++i) out[i] = Int16ToUnitScalar<float>(in[i]); }
float ref_log_spectral_distance(const float* a, const float* b, int n) {
  const auto d = LogSpectralDistance(absl::MakeConstSpan(a, (size_t)n), absl::MakeConstSpan(b, (size_t)n));
  return d.has_value() ? *d : -1.f;
}
// cli_example/encoder_main_lib.cc:99-140 / decoder_main_lib.cc:142-222: whole files through LyraEncoder::Create /
// LyraDecoder::Create.  model_path has to hold files named like the reference's assets (lyra_config.h:117-168 probes
// for them and reads lyra_config.binarypb); their content is never read -- the factories above build the components.
int ref_encode_file(const char* wav_path, const char* out_path, int bitrate, int enable_preprocessing, int enable_dtx,
                    const char* model_path) {
  return EncodeFile(wav_path, out_path, bitrate, enable_preprocessing != 0, enable_dtx != 0, model_path) ? 1 : 0;
}
int ref_decode_file(const char* encoded_path, const char* out_path, int sample_rate_hz, int bitrate, int randomize_requests,
                    float packet_loss_rate, float average_burst_length, const float* loss_starts,
                    const float* loss_durations, int n_loss, const char* model_path) {
  const PacketLossPattern pattern(std::vector<float>(loss_starts, loss_starts + n_loss),
                                  std::vector<float>(loss_durations, loss_durations + n_loss));
  return DecodeFile(encoded_path, out_path, sample_rate_hz, bitrate, randomize_requests != 0, packet_loss_rate,
                    average_burst_length, pattern, model_path) ? 1 : 0;
}
int ref_convert_num_samples(int n, int from_hz, int to_hz) { return ConvertNumSamplesBetweenSampleRate(n, from_hz, to_hz); }

}  // extern "C"
