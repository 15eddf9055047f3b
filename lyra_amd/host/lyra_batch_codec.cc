// lyra_batch_codec.cc -- see lyra_batch_codec.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_batch_codec.h"

#include <string>

#include "../../include/lyra_hip.h"
#include "glog/logging.h"

namespace chromemedia {
namespace codec {
namespace {

// lyra_config.h:56,131-143 (AreParamsSupported) restricted to what this build covers.
bool ParamsSupported(int sample_rate_hz, int num_channels, int num_streams) {
  if (sample_rate_hz != 8000 && sample_rate_hz != 16000 && sample_rate_hz != 32000 && sample_rate_hz != 48000) {
    LOG(ERROR) << "Sample rate " << sample_rate_hz << " Hz is not supported by codec.";
    return false;
  }
  if (sample_rate_hz != kBatchInternalSampleRateHz) {
    LOG(ERROR) << "Sample rate " << sample_rate_hz << " Hz needs the resampler, which this build does not "
               << "provide; feed " << kBatchInternalSampleRateHz << " Hz audio.";
    return false;
  }
  if (num_channels != 1) {
    LOG(ERROR) << "Number of channels " << num_channels << " is not supported by codec. It needs to be 1.";
    return false;
  }
  if (num_streams < 1) {
    LOG(ERROR) << "num_streams must be positive.";
    return false;
  }
  return true;
}

lyra_hip_ctx* NewContext(const ghc::filesystem::path& model_path, int device, int num_streams) {
  lyra_hip_ctx* ctx = nullptr;
  if (lyra_hip_create(model_path.string().c_str(), device, num_streams, LYRA_HIP_REQUANT_EXACT, &ctx) != 0) {
    LOG(ERROR) << "lyra_hip_create failed: " << lyra_hip_last_error(nullptr);
    return nullptr;
  }
  return ctx;
}

}  // namespace

int BatchBitrateToNumQuantizedBits(int bitrate) {
  switch (bitrate) {  // GetBitrate(bits) = ceil(bits / 8) * 8 * 50  (lyra_config.h:79-91)
    case 3200: return 64;
    case 6000: return 120;
    case 9200: return 184;
    default: return -1;
  }
}

int BatchBitrateToPacketSize(int bitrate) { return (bitrate + kBatchFrameRate * 8 - 1) / (kBatchFrameRate * 8); }

static int PacketSizeToBits(int packet_size) {  // PacketSizeToNumQuantizedBits, lyra_config.h:99-106
  switch (packet_size) {
    case 8: return 64;
    case 15: return 120;
    case 23: return 184;
    default: return -1;
  }
}

static std::vector<int32_t> Iota(int n) {
  std::vector<int32_t> v(n);
  for (int i = 0; i < n; ++i) v[i] = i;
  return v;
}

// ---- encoder ---------------------------------------------------------------------------------------------------
BatchLyraEncoder::BatchLyraEncoder(lyra_hip_ctx* ctx, int bitrate, int num_streams)
    : ctx_(ctx), bitrate_(bitrate), num_streams_(num_streams), ids_(Iota(num_streams)) {}
BatchLyraDecoder::BatchLyraDecoder(lyra_hip_ctx* ctx, int num_streams)
    : ctx_(ctx), num_streams_(num_streams), ids_(Iota(num_streams)) {}

std::unique_ptr<BatchLyraEncoder> BatchLyraEncoder::Create(int sample_rate_hz, int num_channels, int bitrate,
                                                            bool enable_dtx,
                                                            const ghc::filesystem::path& model_path,
                                                            int num_streams, int device) {
  if (!ParamsSupported(sample_rate_hz, num_channels, num_streams)) return nullptr;
  if (BatchBitrateToNumQuantizedBits(bitrate) < 0) {
    LOG(ERROR) << "Bitrate " << bitrate << " bps is not supported by codec.";
    return nullptr;
  }
  if (enable_dtx) {
    LOG(ERROR) << "Discontinuous transmission needs the noise estimator, which this build does not provide.";
    return nullptr;
  }
  lyra_hip_ctx* ctx = NewContext(model_path, device, num_streams);
  if (ctx == nullptr) {
    LOG(ERROR) << "Could not create Features Extractor.";
    return nullptr;
  }
  return std::unique_ptr<BatchLyraEncoder>(new BatchLyraEncoder(ctx, bitrate, num_streams));
}

BatchLyraEncoder::~BatchLyraEncoder() { lyra_hip_destroy(ctx_); }

std::optional<std::vector<uint8_t>> BatchLyraEncoder::Encode(const absl::Span<const int16_t> audio) {
  const size_t expected = static_cast<size_t>(num_streams_) * kBatchHopSamples;
  if (audio.size() != expected) {
    LOG(ERROR) << "The number of audio samples has to be exactly " << expected << " (" << num_streams_
               << " streams x " << kBatchHopSamples << "), but is " << audio.size() << ".";
    return std::nullopt;
  }
  std::vector<uint8_t> packets(static_cast<size_t>(num_streams_) * packet_size());
  const int rc = lyra_hip_encode(ctx_, ids_.data(), num_streams_, audio.data(), BatchBitrateToNumQuantizedBits(bitrate_),
                                 packets.data());
  if (rc != 0) {
    LOG(ERROR) << "Unable to extract and quantize features from audio: " << lyra_hip_last_error(ctx_);
    return std::nullopt;
  }
  return packets;
}

bool BatchLyraEncoder::set_bitrate(int bitrate) {
  if (BatchBitrateToNumQuantizedBits(bitrate) < 0) {
    LOG(ERROR) << "Bitrate " << bitrate << " bps is not supported by codec.";
    return false;
  }
  bitrate_ = bitrate;
  return true;
}

// ---- decoder ---------------------------------------------------------------------------------------------------
std::unique_ptr<BatchLyraDecoder> BatchLyraDecoder::Create(int sample_rate_hz, int num_channels,
                                                            const ghc::filesystem::path& model_path,
                                                            int num_streams, int device) {
  if (!ParamsSupported(sample_rate_hz, num_channels, num_streams)) return nullptr;
  lyra_hip_ctx* ctx = NewContext(model_path, device, num_streams);
  if (ctx == nullptr) {
    LOG(ERROR) << "New model could not be instantiated.";
    return nullptr;
  }
  return std::unique_ptr<BatchLyraDecoder>(new BatchLyraDecoder(ctx, num_streams));
}

BatchLyraDecoder::~BatchLyraDecoder() { lyra_hip_destroy(ctx_); }

bool BatchLyraDecoder::SetEncodedPackets(absl::Span<const uint8_t> encoded) {
  if (encoded.size() % num_streams_ != 0 || PacketSizeToBits(static_cast<int>(encoded.size() / num_streams_)) < 0) {
    LOG(ERROR) << "The packet size (" << encoded.size() << " bytes for " << num_streams_
               << " streams) is not supported.";
    return false;
  }
  if (!pending_.empty()) {
    LOG(ERROR) << "Could not add received features to generative model.";  // one hop of features at a time
    return false;
  }
  pending_.assign(encoded.begin(), encoded.end());
  pending_bits_ = PacketSizeToBits(static_cast<int>(encoded.size() / num_streams_));
  return true;
}

std::optional<std::vector<int16_t>> BatchLyraDecoder::DecodeSamples(int num_samples) {
  if (num_samples < 0) {
    LOG(ERROR) << "Number of samples has to be non-negative.";
    return std::nullopt;
  }
  if (num_samples == 0) return std::vector<int16_t>();
  if (next_sample_in_hop_ == kBatchHopSamples) {  // a new hop is needed: run the model on the pending packets
    if (pending_.empty()) {
      LOG(ERROR) << "No packet to decode: packet-loss concealment is not part of this build.";
      return std::nullopt;
    }
    hop_.resize(static_cast<size_t>(num_streams_) * kBatchHopSamples);
    const int rc = lyra_hip_decode(ctx_, ids_.data(), num_streams_, pending_.data(), pending_bits_, hop_.data());
    pending_.clear();
    if (rc != 0) {
      LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx_);
      return std::nullopt;
    }
    next_sample_in_hop_ = 0;
  }
  if (next_sample_in_hop_ + num_samples > kBatchHopSamples) {
    LOG(ERROR) << "Requested " << num_samples << " samples but only " << (kBatchHopSamples - next_sample_in_hop_)
               << " are left in the current hop.";
    return std::nullopt;
  }
  std::vector<int16_t> out(static_cast<size_t>(num_streams_) * num_samples);
  for (int s = 0; s < num_streams_; ++s)
    for (int i = 0; i < num_samples; ++i)
      out[static_cast<size_t>(s) * num_samples + i] = hop_[static_cast<size_t>(s) * kBatchHopSamples + next_sample_in_hop_ + i];
  next_sample_in_hop_ += num_samples;
  return out;
}

}  // namespace codec
}  // namespace chromemedia
