// lyra_batch_codec.h -- batched twins of the reference's public codec classes:
// BatchLyraEncoder mirrors LyraEncoder (lyra/lyra_encoder.h:61-101), BatchLyraDecoder mirrors LyraDecoder
// (lyra/lyra_decoder.h:54-94), for `num_streams` independent streams that advance in lock-step, one 20 ms frame
// per call, through the fused C ABI (lyra_hip_encode / lyra_hip_decode).  Same method names, argument meaning and
// error behaviour (nullptr / nullopt / false + LOG(ERROR), no exceptions); buffers are stream-major:
// audio[s * 320 + i], packets[s * packet_size + j].
//
// Scope: the steady-state path only.  What the reference handles around it on the host is rejected
// at Create(), loudly, instead of being approximated: sample rates other than 16 kHz (resampler), DTX (noise
// estimator decision), and -- in the decoder -- requests for samples without a packet (packet-loss concealment /
// comfort noise; DecodeSamples returns nullopt where LyraDecoder would conceal).
#ifndef LYRA_AMD_HOST_LYRA_BATCH_CODEC_H_
#define LYRA_AMD_HOST_LYRA_BATCH_CODEC_H_
#include <cstdint>
#include <memory>
#include <optional>
#include <vector>

#include "absl/types/span.h"
#include "include/ghc/filesystem.hpp"

struct lyra_hip_ctx;

namespace chromemedia {
namespace codec {

// lyra_config.h:70-168: 16 kHz internal rate, 50 frames/s, 320 samples per hop; lyra_config.cc:44-48 bitrates
constexpr int kBatchInternalSampleRateHz = 16000;
constexpr int kBatchFrameRate = 50;
constexpr int kBatchHopSamples = 320;
// 3200 / 6000 / 9200 bps -> 64 / 120 / 184 bits; -1 if unsupported (BitrateToNumQuantizedBits, lyra_config.cc)
int BatchBitrateToNumQuantizedBits(int bitrate);
// 8 / 15 / 23 bytes (BitrateToPacketSize, lyra_config.cc)
int BatchBitrateToPacketSize(int bitrate);

class BatchLyraEncoder {
 public:
  // Arguments of LyraEncoder::Create (lyra_encoder.h:61-63) + the number of streams.  nullptr if a parameter is
  // unsupported (lyra_encoder.cc:46-66) or outside this build's scope (see above).
  static std::unique_ptr<BatchLyraEncoder> Create(int sample_rate_hz, int num_channels, int bitrate, bool enable_dtx,
                                                  const ghc::filesystem::path& model_path, int num_streams,
                                                  int device = 0);
  ~BatchLyraEncoder();
  // One 20 ms frame of every stream: audio.size() must be num_streams * 320 (lyra_encoder.cc:124-129), else nullopt.
  // Returns num_streams packets of packet_size() bytes each.
  std::optional<std::vector<uint8_t>> Encode(const absl::Span<const int16_t> audio);
  bool set_bitrate(int bitrate);   // lyra_encoder.cc:158-166
  int sample_rate_hz() const { return kBatchInternalSampleRateHz; }
  int num_channels() const { return 1; }
  int bitrate() const { return bitrate_; }
  int frame_rate() const { return kBatchFrameRate; }
  int num_streams() const { return num_streams_; }
  int packet_size() const { return BatchBitrateToPacketSize(bitrate_); }

 private:
  BatchLyraEncoder(lyra_hip_ctx* ctx, int bitrate, int num_streams);
  lyra_hip_ctx* ctx_;
  int bitrate_;
  int num_streams_;
  std::vector<int32_t> ids_;          // stream slots 0 .. num_streams-1 of the context
};

class BatchLyraDecoder {
 public:
  // Arguments of LyraDecoder::Create (lyra_decoder.h:54-56) + the number of streams.
  static std::unique_ptr<BatchLyraDecoder> Create(int sample_rate_hz, int num_channels,
                                                  const ghc::filesystem::path& model_path, int num_streams,
                                                  int device = 0);
  ~BatchLyraDecoder();
  // One packet per stream, all of the same size (8 / 15 / 23 bytes selects the bitrate as
  // PacketSizeToNumQuantizedBits does, lyra_decoder.cc:172-196).  False if the size is not a valid packet size or
  // the previous packets have not been fully decoded yet (the reference queues at most one pending packet's
  // worth of features ahead of the generative model: generative_model_interface.h:50-62).
  bool SetEncodedPackets(absl::Span<const uint8_t> encoded);
  // num_samples <= samples left in the current hop (never straddles a hop, generative_model_interface.h:64-101);
  // 0 returns an empty vector without running the model.  Returns num_streams * num_samples samples.
  std::optional<std::vector<int16_t>> DecodeSamples(int num_samples);
  int sample_rate_hz() const { return kBatchInternalSampleRateHz; }
  int num_channels() const { return 1; }
  int frame_rate() const { return kBatchFrameRate; }
  bool is_comfort_noise() const { return false; }   // no concealment / comfort noise in this build
  int num_streams() const { return num_streams_; }

 private:
  BatchLyraDecoder(lyra_hip_ctx* ctx, int num_streams);
  lyra_hip_ctx* ctx_;
  int num_streams_;
  std::vector<int32_t> ids_;
  std::vector<uint8_t> pending_;      // packets set but not yet decoded
  int pending_bits_ = 0;
  std::vector<int16_t> hop_;          // decoded hop, stream-major [num_streams][320]
  int next_sample_in_hop_ = kBatchHopSamples;  // == 320: nothing decoded is waiting
};

}  // namespace codec
}  // namespace chromemedia
#endif
