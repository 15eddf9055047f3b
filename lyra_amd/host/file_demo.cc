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

int main(int argc, char** argv) {
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
