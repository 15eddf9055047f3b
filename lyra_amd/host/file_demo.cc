// file_demo.cc -- batched encoder_main / decoder_main (cli_example/encoder_main.cc, decoder_main.cc):
//   file_demo <model_dir> <bitrate> <out_dir> <a.wav> [<b.wav> ...]
// writes <out_dir>/<stem>.lyra and <out_dir>/<stem>_decoded.wav for every input, all files transcoded together.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "lyra_file_codec.h"

using namespace chromemedia::codec;
namespace fs = ghc::filesystem;

// --selftest-wav <in.wav> <out.wav>: read with ReadWav16, write back with WriteWav16 (CPU only; used by the tests)
static int SelftestWav(const char* in, const char* out) {
  std::vector<int16_t> samples;
  int ch = 0, rate = 0;
  if (!ReadWav16(in, &samples, &ch, &rate)) return 6;
  std::printf("%d %d %zu\n", ch, rate, samples.size());
  return WriteWav16(out, samples, ch, rate) ? 0 : 7;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "--selftest-wav") return SelftestWav(argv[2], argv[3]);
  if (argc < 5) { std::fprintf(stderr, "usage: %s model_dir bitrate out_dir a.wav [b.wav ...]\n", argv[0]); return 2; }
  const fs::path model_dir = argv[1], out_dir = argv[3];
  const int bitrate = std::atoi(argv[2]);
  std::vector<fs::path> wavs, lyras, decoded;
  for (int i = 4; i < argc; ++i) {
    fs::path w = argv[i];
    wavs.push_back(w);
    lyras.push_back(out_dir / (w.stem().string() + ".lyra"));
    decoded.push_back(out_dir / (w.stem().string() + "_decoded.wav"));
  }
  if (EncodeFiles(wavs, lyras, 1234, false, false, model_dir)) return 3;      // unsupported bitrate
  if (EncodeFiles(wavs, lyras, bitrate, false, true, model_dir)) return 3;    // DTX is outside this build
  if (!EncodeFiles(wavs, lyras, bitrate, false, false, model_dir)) return 4;
  if (!DecodeFiles(lyras, decoded, 16000, bitrate, model_dir)) return 5;
  return 0;
}
