// lyra_batch_codec.cc -- see lyra_batch_codec.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_batch_codec.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>

#include "../../include/lyra_hip.h"
#include "glog/logging.h"

namespace chromemedia {
namespace codec {
namespace {

// lyra_config.h:56,131-143 (AreParamsSupported)
bool ParamsSupported(int sample_rate_hz, int num_channels, int num_streams) {
  if (sample_rate_hz != 8000 && sample_rate_hz != 16000 && sample_rate_hz != 32000 && sample_rate_hz != 48000) {
    LOG(ERROR) << "Sample rate " << sample_rate_hz << " Hz is not supported by codec.";
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
  if (lyra_hip_create(model_path.string().c_str(), device, num_streams, LYRA_HIP_REQUANT_DEFAULT, &ctx) != 0) {
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
BatchLyraEncoder::BatchLyraEncoder(lyra_hip_ctx* ctx, int sample_rate_hz, int bitrate, bool enable_dtx, int num_streams)
    : ctx_(ctx), sample_rate_hz_(sample_rate_hz), bitrate_(bitrate), enable_dtx_(enable_dtx), num_streams_(num_streams),
      ids_(Iota(num_streams)), lengths_(num_streams, 0) {}

std::unique_ptr<BatchLyraEncoder> BatchLyraEncoder::Create(int sample_rate_hz, int num_channels, int bitrate,
                                                            bool enable_dtx,
                                                            const ghc::filesystem::path& model_path,
                                                            int num_streams, int device) {
  if (!ParamsSupported(sample_rate_hz, num_channels, num_streams)) return nullptr;
  if (BatchBitrateToNumQuantizedBits(bitrate) < 0) {
    LOG(ERROR) << "Bitrate " << bitrate << " bps is not supported by codec.";
    return nullptr;
  }
  lyra_hip_ctx* ctx = NewContext(model_path, device, num_streams);
  if (ctx == nullptr) {
    LOG(ERROR) << "Could not create Features Extractor.";
    return nullptr;
  }
  return std::unique_ptr<BatchLyraEncoder>(new BatchLyraEncoder(ctx, sample_rate_hz, bitrate, enable_dtx, num_streams));
}

BatchLyraEncoder::~BatchLyraEncoder() { lyra_hip_destroy(ctx_); }

// The blocking form: upload, kernels and download on the extractor's own stream, one synchronise (lyra_hip_encode).  It
// shares no stream with anything, which matters when a BatchLyraDecoder runs beside it on another host thread: the
// pipelined halves below use a second and a third stream of the context (quantizer, upload), and a process's streams
// share a handful of hardware queues.
std::optional<std::vector<uint8_t>> BatchLyraEncoder::Encode(const absl::Span<const int16_t> audio) {
  if (!in_flight_.empty()) {
    LOG(ERROR) << "Encode() while " << in_flight_.size() << " EncodeAsync() hops are in flight: call WaitEncoded() first.";
    return std::nullopt;
  }
  const int hop_external = sample_rate_hz_ / kBatchFrameRate;
  const size_t expected = static_cast<size_t>(num_streams_) * hop_external;
  if (audio.size() != expected) {
    LOG(ERROR) << "The number of audio samples has to be exactly " << expected << " (" << num_streams_
               << " streams x " << hop_external << "), but is " << audio.size() << ".";
    return std::nullopt;
  }
  const int16_t* pcm = audio.data();
  if (sample_rate_hz_ != kBatchInternalSampleRateHz) {   // lyra_encoder.cc:119-122
    resampled_.resize(static_cast<size_t>(num_streams_) * kBatchHopSamples);
    if (lyra_hip_resample(ctx_, LYRA_HIP_SIDE_ENCODER, ids_.data(), num_streams_, audio.data(), hop_external,
                          sample_rate_hz_, kBatchInternalSampleRateHz, resampled_.data()) != 0) {
      LOG(ERROR) << "Could not resample: " << lyra_hip_last_error(ctx_);
      return std::nullopt;
    }
    pcm = resampled_.data();
  }
  const int bits = BatchBitrateToNumQuantizedBits(bitrate_);
  std::vector<uint8_t> packets(static_cast<size_t>(num_streams_) * packet_size());
  int rc;
  if (enable_dtx_) {   // lyra_encoder.cc:131-141; the estimator's time constants follow the EXTERNAL rate (:82-85)
    rc = lyra_hip_set_encoder_sample_rate(ctx_, sample_rate_hz_);
    if (rc == 0) rc = lyra_hip_encode_dtx(ctx_, ids_.data(), num_streams_, pcm, bits, packets.data(), lengths_.data());
  } else {
    rc = lyra_hip_encode(ctx_, ids_.data(), num_streams_, pcm, bits, packets.data());
    std::fill(lengths_.begin(), lengths_.end(), packet_size());
  }
  if (rc != 0) {
    LOG(ERROR) << "Unable to extract and quantize features from audio: " << lyra_hip_last_error(ctx_);
    return std::nullopt;
  }
  return packets;
}

bool BatchLyraEncoder::EncodeAsync(const absl::Span<const int16_t> audio) {
  const int hop_external = sample_rate_hz_ / kBatchFrameRate;
  const size_t expected = static_cast<size_t>(num_streams_) * hop_external;
  if (audio.size() != expected) {
    LOG(ERROR) << "The number of audio samples has to be exactly " << expected << " (" << num_streams_
               << " streams x " << hop_external << "), but is " << audio.size() << ".";
    return false;
  }
  if (in_flight_.size() >= 2) {
    LOG(ERROR) << "Two hops are already in flight: call WaitEncoded() first.";
    return false;
  }
  const int bits = BatchBitrateToNumQuantizedBits(bitrate_);
  // lyra_encoder.cc:119-156 on the device: the encoder's own resampler (external rate -> 16 kHz), with DTX the noise
  // decision (:131-141; the estimator's time constants follow the EXTERNAL rate, :82-85), extract, quantize, pack
  int rc = enable_dtx_ ? lyra_hip_set_encoder_sample_rate(ctx_, sample_rate_hz_) : 0;
  if (rc == 0)
    rc = lyra_hip_encode_begin(ctx_, ids_.data(), num_streams_, audio.data(), sample_rate_hz_, bits, enable_dtx_ ? 1 : 0);
  if (rc != 0) {
    LOG(ERROR) << "Unable to extract and quantize features from audio: " << lyra_hip_last_error(ctx_);
    return false;
  }
  in_flight_.push_back(packet_size());
  return true;
}

std::optional<std::vector<uint8_t>> BatchLyraEncoder::WaitEncoded() {
  if (in_flight_.empty()) {
    LOG(ERROR) << "WaitEncoded() without a hop in flight.";
    return std::nullopt;
  }
  const int ps = in_flight_.front();
  in_flight_.erase(in_flight_.begin());
  std::vector<uint8_t> packets(static_cast<size_t>(num_streams_) * ps);
  if (lyra_hip_encode_end(ctx_, packets.data(), lengths_.data()) != 0) {
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
namespace {
constexpr int kConcealmentDurationSamples = 1280;   // 0.08 s (lyra_decoder.cc:41-51)
constexpr int kFadeDurationSamples = 640;           // 0.04 s (lyra_decoder.cc:53-62)
constexpr int kFadeToCNG = 1, kFadeFromCNG = -1;    // lyra_decoder.h:98-101

// lyra_decoder.cc:65-91
int GetNumSamplesToGenerate(int num_samples_requested, int samples_generated_so_far, int concealment_progress,
                            int model_samples_available, int cng_samples_available) {
  int samples_remaining_packet;
  if (concealment_progress < 0) samples_remaining_packet = std::abs(concealment_progress);
  else if (concealment_progress < kConcealmentDurationSamples) samples_remaining_packet = model_samples_available % kBatchHopSamples;
  else samples_remaining_packet = cng_samples_available;
  if (samples_remaining_packet == 0) samples_remaining_packet = kBatchHopSamples;
  return std::min(num_samples_requested - samples_generated_so_far, samples_remaining_packet);
}
}  // namespace

BatchLyraDecoder::BatchLyraDecoder(lyra_hip_ctx* ctx, int sample_rate_hz, int num_streams)
    : ctx_(ctx), sample_rate_hz_(sample_rate_hz), num_streams_(num_streams), streams_(num_streams),
      all_ids_(Iota(num_streams)), leftover_(num_streams) {}

std::unique_ptr<BatchLyraDecoder> BatchLyraDecoder::Create(int sample_rate_hz, int num_channels,
                                                            const ghc::filesystem::path& model_path,
                                                            int num_streams, int device) {
  if (!ParamsSupported(sample_rate_hz, num_channels, num_streams)) return nullptr;
  lyra_hip_ctx* ctx = NewContext(model_path, device, num_streams);
  if (ctx == nullptr) {
    LOG(ERROR) << "New model could not be instantiated.";
    return nullptr;
  }
  // DecodeSamples blocks until its samples are on the host: next to an encoder context on another thread the decode
  // kernels should win the arbitration (include/lyra_hip.h lyra_hip_set_stream_priorities)
  if (lyra_hip_set_stream_priorities(ctx, 0, 2, 2) != 0) LOG(WARNING) << "stream priorities: " << lyra_hip_last_error(ctx);
  return std::unique_ptr<BatchLyraDecoder>(new BatchLyraDecoder(ctx, sample_rate_hz, num_streams));
}

BatchLyraDecoder::~BatchLyraDecoder() { lyra_hip_destroy(ctx_); }

bool BatchLyraDecoder::is_comfort_noise(int stream) const {
  return stream >= 0 && stream < num_streams_ && streams_[stream].fade_progress == kFadeDurationSamples;
}

bool BatchLyraDecoder::SetEncodedPackets(absl::Span<const uint8_t> encoded) {
  return SetEncodedPackets(absl::MakeConstSpan(all_ids_), encoded);
}

bool BatchLyraDecoder::SetEncodedPackets(absl::Span<const int32_t> streams, absl::Span<const uint8_t> encoded) {
  if (failed_) {
    LOG(ERROR) << "This decoder failed in the middle of a request; its streams are out of step with the device. Create a new one.";
    return false;
  }
  if (streams.empty()) return encoded.empty();
  const int packet_size = static_cast<int>(encoded.size() / streams.size());
  const int bits = PacketSizeToBits(packet_size);
  if (encoded.size() % streams.size() != 0 || bits < 0) {
    LOG(ERROR) << "The packet size (" << encoded.size() << " bytes for " << streams.size()
               << " streams) is not supported.";
    return false;
  }
  for (int32_t id : streams)
    if (id < 0 || id >= num_streams_) {
      LOG(ERROR) << "Stream " << id << " does not exist.";
      return false;
    }
  for (size_t i = 0; i < streams.size(); ++i) {
    Stream& st = streams_[streams[i]];
    // Finish playing out any concealment or comfort noise packets before moving on to the packet we are receiving
    // (lyra_decoder.cc:187-196).
    if (st.concealment_progress == kConcealmentDurationSamples) st.concealment_progress = -cng_available(st);
    else if (st.concealment_progress > 0) st.concealment_progress = -gan_available(st);
    Entry e;
    e.estimated = false;
    e.bits = bits;
    std::memcpy(e.packet, encoded.data() + i * packet_size, static_cast<size_t>(packet_size));
    st.queue.push_back(e);   // DecodeToLossyFeatures + AddFeatures: done on the device when the hop starts
  }
  return true;
}

std::optional<std::vector<int16_t>> BatchLyraDecoder::DecodeSamples(int num_samples) {
  if (num_samples < 0) {
    LOG(ERROR) << "Number of samples has to be non-negative.";
    return std::nullopt;
  }
  std::vector<int16_t> out(static_cast<size_t>(num_streams_) * num_samples);
  if (!DecodeSamples(num_samples, absl::Span<int16_t>(out.data(), out.size()))) return std::nullopt;
  return out;
}

bool BatchLyraDecoder::DecodeSamples(int num_samples, absl::Span<int16_t> out) {
  if (num_samples < 0) {
    LOG(ERROR) << "Number of samples has to be non-negative.";
    return false;
  }
  if (failed_) {
    LOG(ERROR) << "This decoder failed in the middle of a request; its streams are out of step with the device. Create a new one.";
    return false;
  }
  if (out.size() != static_cast<size_t>(num_streams_) * num_samples) {
    LOG(ERROR) << "Output span has " << out.size() << " samples, expected " << static_cast<size_t>(num_streams_) * num_samples;
    return false;
  }
  // BufferedResampler::FilterAndBuffer (buffered_resampler.cc:63-147); all streams are asked for the same number of
  // samples every time, so their leftover buffers have the same length.  At 16 kHz there is no resampler and no buffer.
  if (!pending_.empty()) {
    LOG(ERROR) << "DecodeSamples() while " << pending_.size() << " DecodeSamplesAsync() requests are in flight: call WaitDecoded() first.";
    return false;
  }
  const bool resampling = sample_rate_hz_ != kBatchInternalSampleRateHz;
  const int leftover = resampling ? static_cast<int>(leftover_[0].size()) : 0;
  const int used = std::min(leftover, num_samples);
  int internal = num_samples;
  if (resampling) {
    internal = 0;
    if (num_samples > leftover) {
      const float ratio = static_cast<float>(sample_rate_hz_) / static_cast<float>(kBatchInternalSampleRateHz);
      internal = static_cast<int>(std::ceil(static_cast<float>(num_samples - leftover) / ratio));
    }
  }
  if (!EnqueueInternal(internal)) {
    (void)lyra_hip_twin_fetch(ctx_, num_streams_, 0, sample_rate_hz_, nullptr);   // abandon the half-assembled request
    // EnqueueInternal does a stream's bookkeeping (packet consumed, fade / concealment progress) in the pass that gathers
    // the device calls' arguments: after a failed round the host state of the streams handled so far has advanced while
    // the device never ran that round.  The reference's LyraDecoder has no such window (one stream, one call); here the
    // decoder refuses every further call instead of decoding from a state that no longer matches the device.
    failed_ = true;
    return false;
  }
  const int produced = resampling ? static_cast<int>(static_cast<long>(internal) * sample_rate_hz_ / kBatchInternalSampleRateHz)
                                  : internal;
  // nothing to splice: the device result IS the answer
  const bool direct = used == 0 && produced == num_samples;
  int16_t* dst = out.data();
  if (!direct) {
    external_.resize(static_cast<size_t>(num_streams_) * produced);
    dst = external_.data();
  }
  if (lyra_hip_twin_fetch(ctx_, num_streams_, internal, sample_rate_hz_, dst) != 0) {
    LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx_);
    failed_ = true;   // the request's samples are lost and the streams have moved on
    return false;
  }
  if (resampling) leftover_count_ = leftover - used + (produced - (num_samples - used));
  if (direct) return true;
  const int to_copy = num_samples - used;
  for (int s = 0; s < num_streams_; ++s) {
    int16_t* o = out.data() + static_cast<size_t>(s) * num_samples;
    std::copy(leftover_[s].begin(), leftover_[s].begin() + used, o);
    leftover_[s].erase(leftover_[s].begin(), leftover_[s].begin() + used);
    if (produced > 0) {
      const int16_t* e = &external_[static_cast<size_t>(s) * produced];
      std::copy(e, e + to_copy, o + used);
      leftover_[s].insert(leftover_[s].end(), e + to_copy, e + produced);
    }
  }
  return true;
}

bool BatchLyraDecoder::DecodeSamplesAsync(int num_samples) {
  if (num_samples < 0) {
    LOG(ERROR) << "Number of samples has to be non-negative.";
    return false;
  }
  if (failed_) {
    LOG(ERROR) << "This decoder failed in the middle of a request; its streams are out of step with the device. Create a new one.";
    return false;
  }
  if (pending_.size() >= 2) {
    LOG(ERROR) << "Two requests are already in flight: call WaitDecoded() first.";
    return false;
  }
  // BufferedResampler::FilterAndBuffer (buffered_resampler.cc:63-147); all streams are asked for the same number of
  // samples every time, so their leftover buffers have the same length -- which is all that is needed here: the COUNT of
  // leftovers after every request begun is known before its samples exist.  At 16 kHz there is no resampler and no buffer.
  const bool resampling = sample_rate_hz_ != kBatchInternalSampleRateHz;
  const int leftover = resampling ? leftover_count_ : 0;
  const int used = std::min(leftover, num_samples);
  int internal = num_samples;
  if (resampling) {
    internal = 0;
    if (num_samples > leftover) {
      const float ratio = static_cast<float>(sample_rate_hz_) / static_cast<float>(kBatchInternalSampleRateHz);
      internal = static_cast<int>(std::ceil(static_cast<float>(num_samples - leftover) / ratio));
    }
  }
  if (!EnqueueInternal(internal)) {
    (void)lyra_hip_twin_fetch(ctx_, num_streams_, 0, sample_rate_hz_, nullptr);   // abandon the half-assembled request
    // EnqueueInternal does a stream's bookkeeping (packet consumed, fade / concealment progress) in the pass that gathers
    // the device calls' arguments: after a failed round the host state of the streams handled so far has advanced while
    // the device never ran that round.  The reference's LyraDecoder has no such window (one stream, one call); here the
    // decoder refuses every further call instead of decoding from a state that no longer matches the device.
    failed_ = true;
    return false;
  }
  const int produced = resampling ? static_cast<int>(static_cast<long>(internal) * sample_rate_hz_ / kBatchInternalSampleRateHz)
                                  : internal;
  if (lyra_hip_twin_fetch_begin(ctx_, num_streams_, internal, sample_rate_hz_) != 0) {
    LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx_);
    failed_ = true;   // the request's samples are lost and the streams have moved on
    return false;
  }
  pending_.push_back(Pending{num_samples, used, produced});
  if (resampling) leftover_count_ = leftover - used + (produced - (num_samples - used));
  return true;
}

bool BatchLyraDecoder::WaitDecoded(absl::Span<int16_t> out) {
  if (pending_.empty()) {
    LOG(ERROR) << "WaitDecoded() without a request in flight.";
    return false;
  }
  const Pending rq = pending_.front();
  if (out.size() != static_cast<size_t>(num_streams_) * rq.num_samples) {
    LOG(ERROR) << "Output span has " << out.size() << " samples, expected " << static_cast<size_t>(num_streams_) * rq.num_samples;
    return false;   // (the request stays in flight)
  }
  pending_.erase(pending_.begin());
  // nothing to splice: the device result IS the answer
  const bool direct = rq.used == 0 && rq.produced == rq.num_samples;
  int16_t* dst = out.data();
  if (!direct) {
    external_.resize(static_cast<size_t>(num_streams_) * rq.produced);
    dst = external_.data();
  }
  if (lyra_hip_twin_fetch_end(ctx_, dst) != 0) {
    LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx_);
    failed_ = true;
    return false;
  }
  if (direct) return true;
  const int to_copy = rq.num_samples - rq.used;
  for (int s = 0; s < num_streams_; ++s) {
    int16_t* o = out.data() + static_cast<size_t>(s) * rq.num_samples;
    std::copy(leftover_[s].begin(), leftover_[s].begin() + rq.used, o);
    leftover_[s].erase(leftover_[s].begin(), leftover_[s].begin() + rq.used);
    if (rq.produced > 0) {
      const int16_t* e = &external_[static_cast<size_t>(s) * rq.produced];
      std::copy(e, e + to_copy, o + rq.used);
      leftover_[s].insert(leftover_[s].end(), e + to_copy, e + rq.produced);
    }
  }
  return true;
}

// LyraDecoder::DecodeSamplesInternal (lyra_decoder.cc:228-315) for all streams.  One pass of the reference's while loop
// per stream and round; the host decides (integers), the device computes: per round one conditioning call per group of
// streams that start a hop, one call that cuts / cross-fades every stream's slice into the request's output, one
// noise-estimator call for the received hops that completed.  Nothing synchronises here.
bool BatchLyraDecoder::EnqueueInternal(int n) {
  std::vector<int32_t>(&need_packet)[3] = need_packet_;
  std::vector<int32_t>&need_estimated = need_estimated_, &need_cng = need_cng_, &need_noise = need_noise_;
  std::vector<uint8_t>(&packets)[3] = packets_;
  std::vector<lyra_hip_twin_slice>& slices = slices_;
  static const int kBits[3] = {64, 120, 184};
  // ONE pass over the streams per round (round 5: the state machine, the gathering of the device calls' arguments and
  // GenerateSamples' bookkeeping used to be separate passes over all streams, with a pass in front to find the active ones;
  // at 4,096 streams the passes were most of a request's host time).  Nothing here depends on a device result: a stream's
  // bookkeeping is done as soon as its arguments are gathered.
  for (bool first = true;; first = false) {
    for (auto& v : need_packet) v.clear();
    for (auto& v : packets) v.clear();
    need_estimated.clear(); need_cng.clear(); need_noise.clear(); slices.clear();
    for (int32_t s = 0; s < num_streams_; ++s) {
      Stream& st = streams_[s];
      if (first) st.done = 0;
      if (st.done >= n) continue;
      // ---- the state machine up to the two model calls ----------------------------------------------------------------
      st.n_gen = GetNumSamplesToGenerate(n, st.done, st.concealment_progress, gan_available(st), cng_available(st));
      st.packet_received = gan_available(st) > 0 && st.concealment_progress == 0;
      if (st.packet_received) st.fade_direction = kFadeFromCNG;
      else if (st.concealment_progress == kConcealmentDurationSamples) st.fade_direction = kFadeToCNG;
      else st.concealment_progress += st.n_gen;
      st.cng_n = st.gen_n = st.n_gen;
      st.next_fade = st.fade_progress + st.fade_direction * st.n_gen;
      if (st.fade_direction == kFadeToCNG && st.fade_progress == kFadeDurationSamples) {
        st.next_fade = kFadeDurationSamples;
        st.gen_n = 0;
      } else if (st.fade_direction == kFadeFromCNG && st.fade_progress == 0) {
        st.next_fade = 0;
        st.cng_n = 0;
      }
      if (st.gen_n > 0) {   // RunGenerativeModel (:317-326) + GenerativeModel::GenerateSamples
        if (gan_available(st) == 0) st.queue.push_back(Entry{true, 0, {}});   // feature_estimator_->Estimate(): zeros
        if (st.gen_n > kBatchHopSamples - st.next_in_hop) {
          LOG(ERROR) << "Model could not be run on features.";
          return false;
        }
        if (st.next_in_hop == 0) {   // RunConditioning: this stream starts a hop
          const Entry& e = st.queue.front();
          if (e.estimated) {
            need_estimated.push_back(s);
          } else {
            const int k = e.bits == 64 ? 0 : (e.bits == 120 ? 1 : 2);
            need_packet[k].push_back(s);
            packets[k].insert(packets[k].end(), e.packet, e.packet + (kBits[k] + 7) / 8);
          }
        }
      }
      if (st.cng_n > 0 && cng_available(st) == 0) {   // RunComfortNoiseGenerator (:328-340)
        need_cng.push_back(s);
        st.cng_has_hop = true;
        st.cng_next = 0;
      }
      // RunModel slices + MaybeOverlapAndInsert (:342-373), and the noise estimator's share (:304-311): a received hop
      // is handed over when its last sample has been generated
      lyra_hip_twin_slice sl;
      sl.id = s;
      sl.gan_off = st.next_in_hop; sl.gen_n = st.gen_n;
      sl.cng_off = st.cng_next; sl.cng_n = st.cng_n;
      sl.fade = st.fade_progress; sl.fade_dir = st.fade_direction;
      sl.out_off = st.done;
      sl.noise_row = -1;
      if (st.packet_received && st.next_in_hop + st.gen_n == kBatchHopSamples) {
        sl.noise_row = static_cast<int32_t>(need_noise.size());
        need_noise.push_back(s);
      }
      slices.push_back(sl);
      // ---- bookkeeping of GenerativeModel::GenerateSamples (generative_model_interface.h:88-101) ----------------------
      if (st.gen_n > 0) {
        st.next_in_hop += st.gen_n;
        if (st.next_in_hop == kBatchHopSamples) { st.next_in_hop = 0; st.queue.pop_front(); }
      }
      if (st.cng_n > 0) {
        st.cng_next += st.cng_n;
        if (st.cng_next == kBatchHopSamples) { st.cng_next = 0; st.cng_has_hop = false; }
      }
      st.fade_progress = st.next_fade;
      st.done += st.n_gen;
    }
    if (slices.empty()) break;   // every stream has its n samples
    // ---- device: RunConditioning of every stream that starts a hop ----------------------------------------------------
    for (int k = 0; k < 3; ++k) {
      const std::vector<int32_t>& ids = need_packet[k];
      if (ids.empty()) continue;
      if (lyra_hip_twin_decode(ctx_, ids.data(), static_cast<int>(ids.size()), packets[k].data(), kBits[k]) != 0) {
        LOG(ERROR) << "Model could not be run on features: " << lyra_hip_last_error(ctx_);
        return false;
      }
    }
    if (!need_estimated.empty() &&
        lyra_hip_twin_conceal(ctx_, need_estimated.data(), static_cast<int>(need_estimated.size())) != 0) {
      LOG(ERROR) << "Could not add estimated features to generative model: " << lyra_hip_last_error(ctx_);
      return false;
    }
    // AddFeatures(noise_estimator_->noise_estimate()) + conditioning: the device reads the estimate in place, BEFORE
    // this round's noise-estimator update, as the reference's statement order has it
    if (!need_cng.empty() && lyra_hip_twin_comfort_noise(ctx_, need_cng.data(), static_cast<int>(need_cng.size())) != 0) {
      LOG(ERROR) << "Could not generate comfort noise: " << lyra_hip_last_error(ctx_);
      return false;
    }
    if (lyra_hip_twin_assemble(ctx_, slices.data(), static_cast<int>(slices.size()), n) != 0) {
      LOG(ERROR) << "Could not overlap comfort noise: " << lyra_hip_last_error(ctx_);
      return false;
    }
    if (!need_noise.empty() && lyra_hip_twin_noise(ctx_, need_noise.data(), static_cast<int>(need_noise.size())) != 0) {
      LOG(ERROR) << "Could not update noise estimator on decoder output: " << lyra_hip_last_error(ctx_);
      return false;
    }
  }
  return true;
}

}  // namespace codec
}  // namespace chromemedia
