// SHADOW of lyra/wav_utils.h (the reference's wav_utils.cc sits on un-vendored audio_dsp/portable/read_wav_file.h): the
// same two functions (wav_utils.h:30-44) over a minimal RIFF/WAVE reader and writer for 16-bit PCM.
#ifndef REF_SHADOW_WAV_UTILS_H_
#define REF_SHADOW_WAV_UTILS_H_
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "absl/status/status.h"
#include "absl/status/statusor.h"

namespace chromemedia::codec {

struct ReadWavResult {
  const std::vector<int16_t> samples;
  const int num_channels;
  const int sample_rate_hz;
};

inline absl::StatusOr<ReadWavResult> Read16BitWavFileToVector(const std::string& file_name) {
  std::ifstream in(file_name, std::ios::binary);
  if (!in) return absl::UnknownError("could not open " + file_name);
  std::vector<uint8_t> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (b.size() < 12 || std::memcmp(b.data(), "RIFF", 4) != 0 || std::memcmp(b.data() + 8, "WAVE", 4) != 0)
    return absl::InvalidArgumentError("not a RIFF/WAVE file: " + file_name);
  int channels = 0, rate = 0, bits = 0;
  std::vector<int16_t> samples;
  size_t p = 12;
  while (p + 8 <= b.size()) {
    uint32_t n;
    std::memcpy(&n, &b[p + 4], 4);
    const size_t body = p + 8, len = body + n <= b.size() ? n : b.size() - body;
    if (std::memcmp(&b[p], "fmt ", 4) == 0 && len >= 16) {
      uint16_t fmt, ch, bps;
      uint32_t sr;
      std::memcpy(&fmt, &b[body], 2); std::memcpy(&ch, &b[body + 2], 2);
      std::memcpy(&sr, &b[body + 4], 4); std::memcpy(&bps, &b[body + 14], 2);
      if (fmt != 1) return absl::InvalidArgumentError("not PCM: " + file_name);
      channels = ch; rate = (int)sr; bits = bps;
    } else if (std::memcmp(&b[p], "data", 4) == 0) {
      samples.resize(len / 2);
      std::memcpy(samples.data(), &b[body], samples.size() * 2);
    }
    p = body + n + (n & 1);
  }
  if (bits != 16 || channels <= 0) return absl::InvalidArgumentError("not 16-bit PCM: " + file_name);
  return ReadWavResult{samples, channels, rate};
}

inline absl::Status Write16BitWavFileFromVector(const std::string& file_name, int num_channels, int sample_rate_hz,
                                                const std::vector<int16_t>& samples) {
  std::ofstream out(file_name, std::ios::binary);
  if (!out) return absl::UnknownError("could not open " + file_name);
  const uint32_t data = (uint32_t)samples.size() * 2, riff = 36 + data, sr = (uint32_t)sample_rate_hz;
  const uint16_t fmt = 1, ch = (uint16_t)num_channels, bps = 16, align = (uint16_t)(num_channels * 2);
  const uint32_t fmt_len = 16, byte_rate = sr * align;
  out.write("RIFF", 4); out.write((const char*)&riff, 4); out.write("WAVEfmt ", 8); out.write((const char*)&fmt_len, 4);
  out.write((const char*)&fmt, 2); out.write((const char*)&ch, 2); out.write((const char*)&sr, 4);
  out.write((const char*)&byte_rate, 4); out.write((const char*)&align, 2); out.write((const char*)&bps, 2);
  out.write("data", 4); out.write((const char*)&data, 4);
  out.write((const char*)samples.data(), data);
  return out ? absl::OkStatus() : absl::UnknownError("write failed: " + file_name);
}

}  // namespace chromemedia::codec
#endif
