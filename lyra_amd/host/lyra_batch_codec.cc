// lyra_batch_codec.cc -- see lyra_batch_codec.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_batch_codec.h"

#include <algorithm>
#include <cmath>
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

std::optional<std::vector<uint8_t>> BatchLyraEncoder::Encode(const absl::Span<const int16_t> audio) {
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
      leftover_(num_streams) {}

std::unique_ptr<BatchLyraDecoder> BatchLyraDecoder::Create(int sample_rate_hz, int num_channels,
                                                            const ghc::filesystem::path& model_path,
                                                            int num_streams, int device) {
  if (!ParamsSupported(sample_rate_hz, num_channels, num_streams)) return nullptr;
  lyra_hip_ctx* ctx = NewContext(model_path, device, num_streams);
  if (ctx == nullptr) {
    LOG(ERROR) << "New model could not be instantiated.";
    return nullptr;
  }
  return std::unique_ptr<BatchLyraDecoder>(new BatchLyraDecoder(ctx, sample_rate_hz, num_streams));
}

BatchLyraDecoder::~BatchLyraDecoder() { lyra_hip_destroy(ctx_); }

bool BatchLyraDecoder::is_comfort_noise(int stream) const {
  return stream >= 0 && stream < num_streams_ && streams_[stream].fade_progress == kFadeDurationSamples;
}

bool BatchLyraDecoder::SetEncodedPackets(absl::Span<const uint8_t> encoded) {
  const std::vector<int32_t> all = Iota(num_streams_);
  return SetEncodedPackets(absl::MakeConstSpan(all), encoded);
}

bool BatchLyraDecoder::SetEncodedPackets(absl::Span<const int32_t> streams, absl::Span<const uint8_t> encoded) {
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
    e.packet.assign(encoded.begin() + i * packet_size, encoded.begin() + (i + 1) * packet_size);
    st.queue.push_back(std::move(e));   // DecodeToLossyFeatures + AddFeatures: done on the device when the hop starts
  }
  return true;
}

std::optional<std::vector<int16_t>> BatchLyraDecoder::DecodeSamples(int num_samples) {
  if (num_samples < 0) {
    LOG(ERROR) << "Number of samples has to be non-negative.";
    return std::nullopt;
  }
  if (sample_rate_hz_ == kBatchInternalSampleRateHz) return DecodeInternal(num_samples);
  // BufferedResampler::FilterAndBuffer (buffered_resampler.cc:63-147); all streams are asked for the same number of
  // samples every time, so their leftover buffers have the same length.
  const int leftover = static_cast<int>(leftover_[0].size());
  const int used = std::min(leftover, num_samples);
  int internal = 0;
  if (num_samples > leftover) {
    const float ratio = static_cast<float>(sample_rate_hz_) / static_cast<float>(kBatchInternalSampleRateHz);
    internal = static_cast<int>(std::ceil(static_cast<float>(num_samples - leftover) / ratio));
  }
  std::vector<int16_t> out(static_cast<size_t>(num_streams_) * num_samples);
  for (int s = 0; s < num_streams_; ++s) {
    std::copy(leftover_[s].begin(), leftover_[s].begin() + used, out.begin() + static_cast<size_t>(s) * num_samples);
    leftover_[s].erase(leftover_[s].begin(), leftover_[s].begin() + used);
  }
  auto internal_samples = DecodeInternal(internal);
  if (!internal_samples.has_value()) return std::nullopt;
  if (internal == 0) return out;
  // the device resampler takes whole multiples of the decimation factor (16 kHz -> 8 kHz: pairs of samples)
  const int down = sample_rate_hz_ < kBatchInternalSampleRateHz ? kBatchInternalSampleRateHz / sample_rate_hz_ : 1;
  if (internal % down != 0 || internal > 960) {
    LOG(ERROR) << "Could not decode samples: " << internal << " internal samples in one request are not supported.";
    return std::nullopt;
  }
  const int produced = internal * sample_rate_hz_ / kBatchInternalSampleRateHz;
  std::vector<int16_t> external(static_cast<size_t>(num_streams_) * produced);
  const std::vector<int32_t> ids = Iota(num_streams_);
  if (lyra_hip_resample(ctx_, LYRA_HIP_SIDE_DECODER, ids.data(), num_streams_, internal_samples->data(), internal,
                        kBatchInternalSampleRateHz, sample_rate_hz_, external.data()) != 0) {
    LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx_);
    return std::nullopt;
  }
  const int to_copy = num_samples - used;
  for (int s = 0; s < num_streams_; ++s) {
    const int16_t* e = &external[static_cast<size_t>(s) * produced];
    std::copy(e, e + to_copy, out.begin() + static_cast<size_t>(s) * num_samples + used);
    leftover_[s].insert(leftover_[s].end(), e + to_copy, e + produced);
  }
  return out;
}

// LyraDecoder::DecodeSamplesInternal (lyra_decoder.cc:228-315) for all streams.  One pass of the reference's while loop
// per stream and round; model runs are gathered per round.
std::optional<std::vector<int16_t>> BatchLyraDecoder::DecodeInternal(int n) {
  for (Stream& st : streams_) { st.out.clear(); st.out.reserve(n); }
  std::vector<int32_t> active, need_packet[3], need_estimated, need_cng, need_noise;
  static const int kBits[3] = {64, 120, 184};
  std::vector<uint8_t> packets;
  std::vector<int16_t> pcm;
  std::vector<float> zeros;
  std::vector<int32_t> flags;
  while (true) {
    active.clear();
    for (int s = 0; s < num_streams_; ++s)
      if (static_cast<int>(streams_[s].out.size()) < n) active.push_back(s);
    if (active.empty()) break;
    for (auto& v : need_packet) v.clear();
    need_estimated.clear(); need_cng.clear(); need_noise.clear();
    // ---- host: the state machine up to the two model calls --------------------------------------------------------
    for (int32_t s : active) {
      Stream& st = streams_[s];
      st.n_gen = GetNumSamplesToGenerate(n, static_cast<int>(st.out.size()), st.concealment_progress, gan_available(st),
                                         cng_available(st));
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
          return std::nullopt;
        }
        if (st.next_in_hop == 0) {
          const Entry& e = st.queue.front();
          if (e.estimated) need_estimated.push_back(s);
          else need_packet[e.bits == 64 ? 0 : (e.bits == 120 ? 1 : 2)].push_back(s);
        }
      }
      if (st.cng_n > 0 && cng_available(st) == 0) need_cng.push_back(s);   // RunComfortNoiseGenerator (:328-340)
    }
    // ---- device: RunConditioning of every stream that starts a hop ----------------------------------------------------
    for (int k = 0; k < 3; ++k) {
      const std::vector<int32_t>& ids = need_packet[k];
      if (ids.empty()) continue;
      const int nb = (kBits[k] + 7) / 8;
      packets.resize(ids.size() * nb);
      for (size_t i = 0; i < ids.size(); ++i)
        std::copy(streams_[ids[i]].queue.front().packet.begin(), streams_[ids[i]].queue.front().packet.end(),
                  packets.begin() + i * nb);
      pcm.resize(ids.size() * kBatchHopSamples);
      if (lyra_hip_decode(ctx_, ids.data(), static_cast<int>(ids.size()), packets.data(), kBits[k], pcm.data()) != 0) {
        LOG(ERROR) << "Model could not be run on features: " << lyra_hip_last_error(ctx_);
        return std::nullopt;
      }
      for (size_t i = 0; i < ids.size(); ++i)
        streams_[ids[i]].hop.assign(pcm.begin() + i * kBatchHopSamples, pcm.begin() + (i + 1) * kBatchHopSamples);
    }
    if (!need_estimated.empty()) {
      zeros.assign(need_estimated.size() * LYRA_HIP_NUM_FEATURES, 0.f);
      pcm.resize(need_estimated.size() * kBatchHopSamples);
      if (lyra_hip_generate(ctx_, need_estimated.data(), static_cast<int>(need_estimated.size()), zeros.data(), pcm.data()) != 0) {
        LOG(ERROR) << "Could not add estimated features to generative model: " << lyra_hip_last_error(ctx_);
        return std::nullopt;
      }
      for (size_t i = 0; i < need_estimated.size(); ++i)
        streams_[need_estimated[i]].hop.assign(pcm.begin() + i * kBatchHopSamples, pcm.begin() + (i + 1) * kBatchHopSamples);
    }
    if (!need_cng.empty()) {
      pcm.resize(need_cng.size() * kBatchHopSamples);
      // AddFeatures(noise_estimator_->noise_estimate()) + conditioning: the device reads the estimate in place
      if (lyra_hip_comfort_noise(ctx_, need_cng.data(), static_cast<int>(need_cng.size()), nullptr, pcm.data()) != 0) {
        LOG(ERROR) << "Could not generate comfort noise: " << lyra_hip_last_error(ctx_);
        return std::nullopt;
      }
      for (size_t i = 0; i < need_cng.size(); ++i) {
        Stream& st = streams_[need_cng[i]];
        st.cng_hop.assign(pcm.begin() + i * kBatchHopSamples, pcm.begin() + (i + 1) * kBatchHopSamples);
        st.cng_has_hop = true;
        st.cng_next = 0;
      }
    }
    // ---- host: RunModel slices, MaybeOverlapAndInsert (:342-373), bookkeeping --------------------------------------------
    for (int32_t s : active) {
      Stream& st = streams_[s];
      const int16_t* audio = st.gen_n > 0 ? &st.hop[st.next_in_hop] : nullptr;
      const int16_t* noise = st.cng_n > 0 ? &st.cng_hop[st.cng_next] : nullptr;
      if (noise == nullptr) {
        st.out.insert(st.out.end(), audio, audio + st.gen_n);
      } else if (audio == nullptr) {
        st.out.insert(st.out.end(), noise, noise + st.cng_n);
      } else {
        int fade = st.fade_progress;
        for (int i = 0; i < st.gen_n; ++i) {
          const float w = (1.f + std::cos(fade * M_PI / kFadeDurationSamples)) / 2.f;
          st.out.push_back(static_cast<int16_t>(audio[i] * w + noise[i] * (1.f - w)));
          fade += st.fade_direction;
        }
      }
      if (st.packet_received) st.noise_in.insert(st.noise_in.end(), audio, audio + st.gen_n);
      if (st.gen_n > 0) {
        st.next_in_hop += st.gen_n;
        if (st.next_in_hop == kBatchHopSamples) { st.next_in_hop = 0; st.queue.pop_front(); }
      }
      if (st.cng_n > 0) {
        st.cng_next += st.cng_n;
        if (st.cng_next == kBatchHopSamples) { st.cng_next = 0; st.cng_has_hop = false; }
      }
      st.fade_progress = st.next_fade;
      if (static_cast<int>(st.noise_in.size()) == kBatchHopSamples) need_noise.push_back(s);
    }
    // ---- device: noise_estimator_->ReceiveSamples for every stream whose received hop is complete (:304-311) ---------
    if (!need_noise.empty()) {
      pcm.resize(need_noise.size() * kBatchHopSamples);
      for (size_t i = 0; i < need_noise.size(); ++i) {
        Stream& st = streams_[need_noise[i]];
        std::copy(st.noise_in.begin(), st.noise_in.end(), pcm.begin() + i * kBatchHopSamples);
        st.noise_in.clear();
      }
      flags.resize(need_noise.size());
      if (lyra_hip_noise_receive(ctx_, LYRA_HIP_SIDE_DECODER, need_noise.data(), static_cast<int>(need_noise.size()),
                                 pcm.data(), flags.data()) != 0) {
        LOG(ERROR) << "Could not update noise estimator on decoder output: " << lyra_hip_last_error(ctx_);
        return std::nullopt;
      }
    }
  }
  std::vector<int16_t> result(static_cast<size_t>(num_streams_) * n);
  for (int s = 0; s < num_streams_; ++s) std::copy(streams_[s].out.begin(), streams_[s].out.end(), result.begin() + static_cast<size_t>(s) * n);
  return result;
}

}  // namespace codec
}  // namespace chromemedia
