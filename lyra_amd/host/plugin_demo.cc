// plugin_demo.cc -- exercises the C++ plugin surface the way LyraEncoder::Encode / LyraDecoder do
// (lyra/lyra_encoder.cc:143-155, lyra/lyra_decoder.cc:198-207,317-326; benchmark loop lyra_benchmark_lib.cc:85-160):
//   plugin_demo <model_dir> <pcm_in.s16> <num_bits> <bits_out.txt> <pcm_out.s16>
// Reads raw int16 PCM (multiple of 320 samples), writes one '0'/'1' line per hop and the decoded PCM.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <fstream>
#include <vector>

#include "lyra_hip_components.h"

using namespace chromemedia::codec;

int main(int argc, char** argv) {
  if (argc != 6) { std::fprintf(stderr, "usage: %s model_dir pcm_in num_bits bits_out pcm_out\n", argv[0]); return 2; }
  const std::string model_dir = argv[1];
  const int num_bits = std::atoi(argv[3]);
  std::ifstream in(argv[2], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<int16_t> pcm(raw.size() / 2);
  std::memcpy(pcm.data(), raw.data(), pcm.size() * 2);
  auto extractor = CreateFeatureExtractor(model_dir);
  auto quantizer = CreateQuantizer(model_dir);
  auto model = CreateGenerativeModel(64, model_dir);
  if (!extractor || !quantizer || !model) { std::fprintf(stderr, "creation failed\n"); return 1; }
  // reference validation behaviour
  if (quantizer->Quantize(std::vector<float>(64, 0.f), 185).has_value()) return 3;
  if (quantizer->Quantize(std::vector<float>(64, 0.f), 62).has_value()) return 3;
  if (model->GenerateSamples(1).has_value()) return 3;
  std::ofstream bits_out(argv[4]);
  std::ofstream pcm_out(argv[5], std::ios::binary);
  for (size_t hop = 0; hop + 320 <= pcm.size(); hop += 320) {
    auto feats = extractor->Extract(absl::MakeConstSpan(pcm.data() + hop, 320));
    if (!feats) return 4;
    auto bits = quantizer->Quantize(*feats, num_bits);
    if (!bits) return 4;
    bits_out << *bits << "\n";
    auto lossy = quantizer->DecodeToLossyFeatures(*bits);
    if (!lossy || !model->AddFeatures(*lossy)) return 4;
    auto a = model->GenerateSamples(100);   // partial requests inside a hop, as LyraDecoder may issue
    auto b = model->GenerateSamples(220);
    if (!a || !b || model->GenerateSamples(1).has_value()) return 5;
    pcm_out.write(reinterpret_cast<const char*>(a->data()), a->size() * 2);
    pcm_out.write(reinterpret_cast<const char*>(b->data()), b->size() * 2);
  }
  return 0;
}
