// lyra_batch_codec.h -- batched twins of the reference's public codec classes:
// BatchLyraEncoder mirrors LyraEncoder (lyra/lyra_encoder.h:61-101), BatchLyraDecoder mirrors LyraDecoder
// (lyra/lyra_decoder.h:54-94), for `num_streams` independent streams served by one GPU context.  Same method names,
// argument meaning and error behaviour (nullptr / nullopt / false + LOG(ERROR), no exceptions); buffers are
// stream-major: audio[s * n + i], packets[s * packet_size + j].
//
// Everything the reference's classes do around the three plugins is here, per stream, with the heavy parts batched
// on the device through the C ABI (include/lyra_hip.h):
//  * sample rates 8 / 16 / 32 / 48 kHz (Resampler on the way in, BufferedResampler's leftover logic on the way out);
//  * DTX: encoder-side NoiseEstimator, empty packets for noise hops (lyra_encoder.cc:131-141);
//  * decoder: per-stream packet FIFO, DecodeSamples(n) for any n (requests may straddle hops), packet-loss
//    concealment with estimated (zero) features, comfort noise from the decoder-side noise estimate, cosine
//    cross-fades, noise-estimator updates on received hops only (lyra_decoder.cc:172-373).
// The control flow is the reference's, statement for statement, run for every stream ON INTEGERS ONLY: the conditioned
// hops of the generative model and of the comfort-noise generator stay on the device (include/lyra_hip.h "Decoder
// twin").  In each round of the decode loop the streams that need a new hop from the generative model / the
// comfort-noise generator / a noise-estimator update are collected and served by ONE device call each, the round's
// slices and cross-fades by one more; nothing synchronises until the request's single device-to-host copy.  Per
// DecodeSamples call: the queued packets and a few integers per stream go up, the result comes down.
#ifndef LYRA_AMD_HOST_LYRA_BATCH_CODEC_H_
#define LYRA_AMD_HOST_LYRA_BATCH_CODEC_H_
#include <cstdint>
#include <deque>
#include <memory>
#include <optional>
#include <vector>

#include "../../include/lyra_hip.h"
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
  // unsupported (lyra_encoder.cc:46-66).
  static std::unique_ptr<BatchLyraEncoder> Create(int sample_rate_hz, int num_channels, int bitrate, bool enable_dtx,
                                                  const ghc::filesystem::path& model_path, int num_streams,
                                                  int device = 0);
  ~BatchLyraEncoder();
  // One 20 ms frame of every stream at sample_rate_hz(): audio.size() must be num_streams * sample_rate_hz / 50
  // (lyra_encoder.cc:124-129), else nullopt.  Returns num_streams rows of packet_size() bytes; with DTX a row whose
  // stream sent an empty packet (lyra_encoder.cc:136-141) is all zero and packet_lengths()[s] == 0.
  std::optional<std::vector<uint8_t>> Encode(const absl::Span<const int16_t> audio);
  // The same in two halves, for a caller that feeds hop after hop (round 6): EncodeAsync() starts a hop and returns --
  // `audio` may be reused at once --, WaitEncoded() returns the packets of the OLDEST hop started.  Up to two hops may be
  // in flight; with EncodeAsync(n + 1) issued before WaitEncoded(n) the upload of hop n + 1 and the download of hop n run
  // under the kernels (include/lyra_hip.h "Pipelined host-buffer calls").  Same packets as Encode().
  bool EncodeAsync(const absl::Span<const int16_t> audio);
  std::optional<std::vector<uint8_t>> WaitEncoded();
  int hops_in_flight() const { return static_cast<int>(in_flight_.size()); }
  // Bytes of each stream's packet from the last Encode / WaitEncoded: packet_size(), or 0 for a DTX empty packet.
  const std::vector<int32_t>& packet_lengths() const { return lengths_; }
  bool set_bitrate(int bitrate);   // lyra_encoder.cc:158-166
  int sample_rate_hz() const { return sample_rate_hz_; }
  int num_channels() const { return 1; }
  int bitrate() const { return bitrate_; }
  int frame_rate() const { return kBatchFrameRate; }
  int num_streams() const { return num_streams_; }
  int packet_size() const { return BatchBitrateToPacketSize(bitrate_); }

 private:
  BatchLyraEncoder(lyra_hip_ctx* ctx, int sample_rate_hz, int bitrate, bool enable_dtx, int num_streams);
  lyra_hip_ctx* ctx_;
  int sample_rate_hz_;
  int bitrate_;
  bool enable_dtx_;
  int num_streams_;
  std::vector<int32_t> ids_;          // stream slots 0 .. num_streams-1 of the context
  std::vector<int32_t> lengths_;
  std::vector<int16_t> resampled_;
  std::vector<int> in_flight_;        // packet size of every hop begun and not yet waited for, oldest first
};

class BatchLyraDecoder {
 public:
  // Arguments of LyraDecoder::Create (lyra_decoder.h:54-56) + the number of streams.
  static std::unique_ptr<BatchLyraDecoder> Create(int sample_rate_hz, int num_channels,
                                                  const ghc::filesystem::path& model_path, int num_streams,
                                                  int device = 0);
  ~BatchLyraDecoder();
  // LyraDecoder::SetEncodedPacket (lyra_decoder.cc:172-209) for every stream: num_streams packets of one size
  // (8 / 15 / 23 bytes selects the bitrate as PacketSizeToNumQuantizedBits does).  Packets queue up per stream.
  bool SetEncodedPackets(absl::Span<const uint8_t> encoded);
  // ... for the listed streams only (a stream that lost its packet this tick is simply not listed).
  bool SetEncodedPackets(absl::Span<const int32_t> streams, absl::Span<const uint8_t> encoded);
  // LyraDecoder::DecodeSamples (lyra_decoder.cc:211-315) for every stream: any num_samples >= 0 at sample_rate_hz();
  // streams without a packet conceal, then fade to comfort noise.  Returns num_streams * num_samples samples.
  std::optional<std::vector<int16_t>> DecodeSamples(int num_samples);
  // The same into caller memory (num_streams * num_samples samples; pinned memory avoids a staging copy): no allocation
  // on the steady-state path.  false + LOG(ERROR) where the other form returns nullopt.
  bool DecodeSamples(int num_samples, absl::Span<int16_t> out);
  // The same in two halves (round 6): DecodeSamplesAsync() runs the state machine and enqueues the request -- packets for
  // the next request may be set and the next request started at once --, WaitDecoded() delivers the OLDEST request begun
  // (out.size() == num_streams * that request's num_samples).  Up to two requests may be in flight; the download of
  // request n then runs under the kernels of request n + 1.  Same samples as DecodeSamples().
  bool DecodeSamplesAsync(int num_samples);
  bool WaitDecoded(absl::Span<int16_t> out);
  int requests_in_flight() const { return static_cast<int>(pending_.size()); }
  int sample_rate_hz() const { return sample_rate_hz_; }
  int num_channels() const { return 1; }
  int frame_rate() const { return kBatchFrameRate; }
  // LyraDecoder::is_comfort_noise (lyra_decoder.h:88-90) of one stream.
  bool is_comfort_noise(int stream) const;
  int num_streams() const { return num_streams_; }

 private:
  BatchLyraDecoder(lyra_hip_ctx* ctx, int sample_rate_hz, int num_streams);
  // GenerativeModel's FIFO bookkeeping (generative_model_interface.h:45-134) without the model.
  // (plain data: a stream gets a packet per hop, and at 4,096 streams a heap block per packet and a deque per stream were
  // most of DecodeSamples' host time -- round 5, profiles/r05_batch_twins_host_time.txt)
  struct Entry { bool estimated; int bits; uint8_t packet[(4 * LYRA_HIP_MAX_STAGES + 7) / 8]; };
  class HopQueue {   // FIFO of the few conditioning inputs a stream holds (usually 0-2): a vector and a head index
   public:
    size_t size() const { return v_.size() - head_; }
    bool empty() const { return head_ == v_.size(); }
    const Entry& front() const { return v_[head_]; }
    void push_back(const Entry& e) { v_.push_back(e); }
    void pop_front() {
      if (++head_ == v_.size()) { v_.clear(); head_ = 0; }
      else if (head_ >= 32 && 2 * head_ >= v_.size()) { v_.erase(v_.begin(), v_.begin() + head_); head_ = 0; }
    }
   private:
    std::vector<Entry> v_;
    size_t head_ = 0;
  };
  struct Stream {
    HopQueue queue;                       // generative model: queued conditioning inputs
    int next_in_hop = 0;                  // generative model: next_sample_in_hop_ (the hop itself lives on the device)
    bool cng_has_hop = false;             // comfort noise generator: one hop at most is ever queued
    int cng_next = 0;
    int concealment_progress = 0;
    int fade_progress = 0;
    int fade_direction = -1;              // kFadeFromCNG (lyra_decoder.h FadeDirection)
    int done = 0;                         // internal-rate samples of the current request produced so far
    // scratch of the current round
    int n_gen = 0, gen_n = 0, cng_n = 0, next_fade = 0;
    bool packet_received = false;
  };
  bool EnqueueInternal(int num_internal_samples);   // DecodeSamplesInternal for all streams, enqueued on the device
  int gan_available(const Stream& s) const { return (int)s.queue.size() * kBatchHopSamples - s.next_in_hop; }
  int cng_available(const Stream& s) const { return s.cng_has_hop ? kBatchHopSamples - s.cng_next : 0; }

  lyra_hip_ctx* ctx_;
  int sample_rate_hz_;
  int num_streams_;
  bool failed_ = false;                          // a device call failed mid-request: every further call is refused
  std::vector<Stream> streams_;
  std::vector<int32_t> all_ids_;                 // 0 .. num_streams - 1
  // scratch of EnqueueInternal (kept across calls: no allocation per request)
  std::vector<int32_t> need_packet_[3], need_estimated_, need_cng_, need_noise_;
  std::vector<uint8_t> packets_[3];
  std::vector<lyra_hip_twin_slice> slices_;
  struct Pending { int num_samples, used, produced; };   // a request begun: samples asked for, taken from the leftovers, fetched
  std::vector<Pending> pending_;                 // oldest first
  int leftover_count_ = 0;                       // leftover samples per stream once every request begun has been delivered
  std::vector<std::vector<int16_t>> leftover_;   // BufferedResampler::leftover_samples_ per stream (same length for all)
  std::vector<int16_t> external_;                // a request's resampled samples when leftovers have to be spliced in
};

}  // namespace codec
}  // namespace chromemedia
#endif
