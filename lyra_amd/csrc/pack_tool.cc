// pack_tool.cc -- offline form of what lyra_hip_create() does with a reference model directory:
//   pack_tool <model_dir with the three .tflite + lyra_config.binarypb> <out.lyrapack>
#include <cstdio>

#include "tflite_pack.h"

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: %s model_dir out.lyrapack\n", argv[0]); return 2; }
  std::vector<uint8_t> image;
  std::string err;
  if (!lyra::pack_from_tflite_dir(argv[1], &image, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  FILE* f = std::fopen(argv[2], "wb");
  if (!f || std::fwrite(image.data(), 1, image.size(), f) != image.size()) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  std::fclose(f);
  std::printf("%zu bytes -> %s\n", image.size(), argv[2]);
  return 0;
}
