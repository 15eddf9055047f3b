// lyra_file_codec.cc -- see lyra_file_codec.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_file_codec.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>

#include "../../include/lyra_hip.h"
#include "glog/logging.h"
#include "lyra_batch_codec.h"

namespace chromemedia {
namespace codec {
namespace {

struct Ctx {  // RAII around one GPU context
  lyra_hip_ctx* c = nullptr;
  ~Ctx() { if (c) lyra_hip_destroy(c); }
};

bool CheckScope(int num_channels, int sample_rate_hz, bool enable_preprocessing, bool enable_dtx) {
  if (num_channels != 1) {
    LOG(ERROR) << "Number of channels " << num_channels << " is not supported by codec. It needs to be 1.";
    return false;
  }
  if (sample_rate_hz != kBatchInternalSampleRateHz) {
    LOG(ERROR) << "Sample rate " << sample_rate_hz << " Hz needs the resampler, which this build does not provide.";
    return false;
  }
  if (enable_preprocessing || enable_dtx) {
    LOG(ERROR) << "Preprocessing / DTX are not part of this build.";
    return false;
  }
  return true;
}

uint32_t Rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t Rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
void Wr32(std::vector<uint8_t>* v, uint32_t x) { for (int i = 0; i < 4; ++i) v->push_back((x >> (8 * i)) & 255); }
void Wr16(std::vector<uint8_t>* v, uint16_t x) { v->push_back(x & 255); v->push_back(x >> 8); }

}  // namespace

bool ReadWav16(const ghc::filesystem::path& path, std::vector<int16_t>* samples, int* num_channels,
               int* sample_rate_hz) {
  std::ifstream in(path.string(), std::ios::binary);
  if (!in.is_open()) { LOG(ERROR) << "Could not open " << path.string(); return false; }
  std::vector<uint8_t> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (b.size() < 12 || std::memcmp(b.data(), "RIFF", 4) != 0 || std::memcmp(b.data() + 8, "WAVE", 4) != 0) {
    LOG(ERROR) << path.string() << " is not a RIFF/WAVE file.";
    return false;
  }
  bool have_fmt = false;
  size_t pos = 12;
  while (pos + 8 <= b.size()) {
    const uint32_t len = Rd32(&b[pos + 4]);
    const uint8_t* body = &b[pos + 8];
    if (std::memcmp(&b[pos], "fmt ", 4) == 0 && len >= 16 && pos + 8 + 16 <= b.size()) {
      const int format = Rd16(body), bits = Rd16(body + 14);
      *num_channels = Rd16(body + 2);
      *sample_rate_hz = (int)Rd32(body + 4);
      if (format != 1 || bits != 16) { LOG(ERROR) << path.string() << ": only 16-bit PCM is supported."; return false; }
      have_fmt = true;
    } else if (std::memcmp(&b[pos], "data", 4) == 0) {
      if (!have_fmt) { LOG(ERROR) << path.string() << ": data chunk before fmt chunk."; return false; }
      const size_t n = std::min<size_t>(len, b.size() - (pos + 8)) / 2;
      samples->resize(n);
      std::memcpy(samples->data(), body, n * 2);
      return true;
    }
    pos += 8 + (size_t)len + (len & 1);
  }
  LOG(ERROR) << path.string() << ": no data chunk.";
  return false;
}

bool WriteWav16(const ghc::filesystem::path& path, const std::vector<int16_t>& samples, int num_channels,
                int sample_rate_hz) {
  std::vector<uint8_t> h;
  const uint32_t data_bytes = (uint32_t)(samples.size() * 2);
  h.insert(h.end(), {'R', 'I', 'F', 'F'}); Wr32(&h, 36 + data_bytes);
  h.insert(h.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '}); Wr32(&h, 16);
  Wr16(&h, 1); Wr16(&h, (uint16_t)num_channels); Wr32(&h, (uint32_t)sample_rate_hz);
  Wr32(&h, (uint32_t)(sample_rate_hz * num_channels * 2)); Wr16(&h, (uint16_t)(num_channels * 2)); Wr16(&h, 16);
  h.insert(h.end(), {'d', 'a', 't', 'a'}); Wr32(&h, data_bytes);
  std::ofstream out(path.string(), std::ios::binary | std::ios::trunc);
  if (!out.is_open()) { LOG(ERROR) << "Could not open output file " << path.string(); return false; }
  out.write(reinterpret_cast<const char*>(h.data()), h.size());
  out.write(reinterpret_cast<const char*>(samples.data()), data_bytes);
  return out.good();
}

bool EncodeWavs(const std::vector<std::vector<int16_t>>& wav_data, int num_channels, int sample_rate_hz, int bitrate,
                bool enable_preprocessing, bool enable_dtx, const ghc::filesystem::path& model_path,
                std::vector<std::vector<uint8_t>>* encoded_features, int device) {
  if (!CheckScope(num_channels, sample_rate_hz, enable_preprocessing, enable_dtx)) return false;
  const int num_bits = BatchBitrateToNumQuantizedBits(bitrate);
  if (num_bits < 0) { LOG(ERROR) << "Bitrate " << bitrate << " bps is not supported by codec."; return false; }
  const int n = (int)wav_data.size();
  encoded_features->assign(n, {});
  if (n == 0) return true;
  Ctx ctx;
  if (lyra_hip_create(model_path.string().c_str(), device, n, LYRA_HIP_REQUANT_DEFAULT, &ctx.c) != 0) {
    LOG(ERROR) << "Could not create lyra encoder: " << lyra_hip_last_error(nullptr);
    return false;
  }
  const int packet_size = BatchBitrateToPacketSize(bitrate);
  size_t max_hops = 0;
  for (const auto& w : wav_data) max_hops = std::max(max_hops, w.size() / kBatchHopSamples);
  std::vector<int32_t> ids;
  std::vector<int16_t> pcm;
  std::vector<uint8_t> packets;
  for (size_t hop = 0; hop < max_hops; ++hop) {  // streams with a full hop left (encoder_main_lib.cc:71-73)
    ids.clear();
    pcm.clear();
    for (int i = 0; i < n; ++i)
      if ((hop + 1) * kBatchHopSamples <= wav_data[i].size()) {
        ids.push_back(i);
        pcm.insert(pcm.end(), wav_data[i].begin() + hop * kBatchHopSamples,
                   wav_data[i].begin() + (hop + 1) * kBatchHopSamples);
      }
    packets.resize(ids.size() * packet_size);
    if (lyra_hip_encode(ctx.c, ids.data(), (int)ids.size(), pcm.data(), num_bits, packets.data()) != 0) {
      LOG(ERROR) << "Unable to encode features starting at samples at byte " << hop * kBatchHopSamples << ": "
                 << lyra_hip_last_error(ctx.c);
      return false;
    }
    for (size_t k = 0; k < ids.size(); ++k) {
      auto& dst = (*encoded_features)[ids[k]];
      dst.insert(dst.end(), packets.begin() + k * packet_size, packets.begin() + (k + 1) * packet_size);
    }
  }
  return true;
}

bool EncodeFiles(const std::vector<ghc::filesystem::path>& wav_paths,
                 const std::vector<ghc::filesystem::path>& output_paths, int bitrate, bool enable_preprocessing,
                 bool enable_dtx, const ghc::filesystem::path& model_path, int device) {
  if (wav_paths.size() != output_paths.size()) { LOG(ERROR) << "One output path per input file is required."; return false; }
  std::vector<std::vector<int16_t>> wavs(wav_paths.size());
  int channels = 1, rate = kBatchInternalSampleRateHz;
  for (size_t i = 0; i < wav_paths.size(); ++i) {
    int ch = 0, sr = 0;
    if (!ReadWav16(wav_paths[i], &wavs[i], &ch, &sr)) return false;
    if (i == 0) { channels = ch; rate = sr; }
    if (ch != channels || sr != rate) { LOG(ERROR) << "All files of a batch must share channels / sample rate."; return false; }
  }
  std::vector<std::vector<uint8_t>> encoded;
  if (!EncodeWavs(wavs, channels, rate, bitrate, enable_preprocessing, enable_dtx, model_path, &encoded, device)) {
    LOG(ERROR) << "Unable to encode features for the batch starting with " << (wav_paths.empty() ? "" : wav_paths[0].string());
    return false;
  }
  for (size_t i = 0; i < output_paths.size(); ++i) {
    std::ofstream out(output_paths[i].string(), std::ios_base::binary | std::ios_base::trunc);
    if (!out.is_open()) { LOG(ERROR) << "Could not open output file " << output_paths[i].string(); return false; }
    out.write(reinterpret_cast<const char*>(encoded[i].data()), encoded[i].size());
  }
  return true;
}

bool DecodeFeaturesBatch(const std::vector<std::vector<uint8_t>>& packet_streams, int packet_size,
                         const ghc::filesystem::path& model_path, std::vector<std::vector<int16_t>>* decoded_audio,
                         int device) {
  int num_bits = -1;
  for (int br : {3200, 6000, 9200})
    if (BatchBitrateToPacketSize(br) == packet_size) num_bits = BatchBitrateToNumQuantizedBits(br);
  if (num_bits < 0) { LOG(ERROR) << "The packet size (" << packet_size << " bytes) is not supported."; return false; }
  const int n = (int)packet_streams.size();
  decoded_audio->assign(n, {});
  if (n == 0) return true;
  size_t max_packets = 0;
  for (const auto& p : packet_streams) {
    if (p.size() % packet_size != 0) { LOG(ERROR) << "Encoded stream is not a whole number of packets."; return false; }
    max_packets = std::max(max_packets, p.size() / packet_size);
  }
  Ctx ctx;
  if (lyra_hip_create(model_path.string().c_str(), device, n, LYRA_HIP_REQUANT_DEFAULT, &ctx.c) != 0) {
    LOG(ERROR) << "Could not create lyra decoder: " << lyra_hip_last_error(nullptr);
    return false;
  }
  std::vector<int32_t> ids;
  std::vector<uint8_t> packets;
  std::vector<int16_t> pcm;
  for (size_t f = 0; f < max_packets; ++f) {
    ids.clear();
    packets.clear();
    for (int i = 0; i < n; ++i)
      if ((f + 1) * packet_size <= packet_streams[i].size()) {
        ids.push_back(i);
        packets.insert(packets.end(), packet_streams[i].begin() + f * packet_size,
                       packet_streams[i].begin() + (f + 1) * packet_size);
      }
    pcm.resize(ids.size() * kBatchHopSamples);
    if (lyra_hip_decode(ctx.c, ids.data(), (int)ids.size(), packets.data(), num_bits, pcm.data()) != 0) {
      LOG(ERROR) << "Could not decode samples: " << lyra_hip_last_error(ctx.c);
      return false;
    }
    for (size_t k = 0; k < ids.size(); ++k) {
      auto& dst = (*decoded_audio)[ids[k]];
      dst.insert(dst.end(), pcm.begin() + k * kBatchHopSamples, pcm.begin() + (k + 1) * kBatchHopSamples);
    }
  }
  return true;
}

bool DecodeFiles(const std::vector<ghc::filesystem::path>& encoded_paths,
                 const std::vector<ghc::filesystem::path>& output_paths, int sample_rate_hz, int bitrate,
                 const ghc::filesystem::path& model_path, int device) {
  if (encoded_paths.size() != output_paths.size()) { LOG(ERROR) << "One output path per input file is required."; return false; }
  if (!CheckScope(1, sample_rate_hz, false, false)) return false;
  if (BatchBitrateToNumQuantizedBits(bitrate) < 0) { LOG(ERROR) << "Bitrate " << bitrate << " bps is not supported by codec."; return false; }
  std::vector<std::vector<uint8_t>> streams(encoded_paths.size());
  for (size_t i = 0; i < encoded_paths.size(); ++i) {
    std::ifstream in(encoded_paths[i].string(), std::ios::binary);
    if (!in.is_open()) { LOG(ERROR) << "Could not open " << encoded_paths[i].string(); return false; }
    streams[i].assign((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  }
  std::vector<std::vector<int16_t>> audio;
  if (!DecodeFeaturesBatch(streams, BatchBitrateToPacketSize(bitrate), model_path, &audio, device)) return false;
  for (size_t i = 0; i < output_paths.size(); ++i)
    if (!WriteWav16(output_paths[i], audio[i], 1, sample_rate_hz)) return false;
  return true;
}

}  // namespace codec
}  // namespace chromemedia
