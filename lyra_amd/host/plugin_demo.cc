// plugin_demo.cc -- exercises the C++ plugin surface the way LyraEncoder::Encode / LyraDecoder do
// (lyra/lyra_encoder.cc:143-155, lyra/lyra_decoder.cc:198-207,317-326; benchmark loop lyra_benchmark_lib.cc:85-160):
//   plugin_demo <model_dir> <pcm_in.s16> <num_bits> <bits_out.txt> <pcm_out.s16>
// Reads raw int16 PCM (multiple of 320 samples), writes one '0'/'1' line per hop and the decoded PCM.
//   plugin_demo --bench <model_dir> [num_cond_vectors = 2000] [num_bits = 120]
// lyra_benchmark at the plugin boundary (lyra_benchmark_lib.cc:85-260): ONE stream, random full-scale hops
// (UnitToInt16Scalar(U(-1,1)), :233-239), per-stage max / min / mean / stdev of Extract, Quantize, DecodeToLossyFeatures,
// AddFeatures + GenerateSamples(320) in the reference's own output format -- the one number directly comparable with
// what the reference publishes (README: 0.153 / 0.130 / 0.029 / 0.212 ms per hop on a Pixel 6).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <fstream>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "lyra_hip_components.h"

using namespace chromemedia::codec;

namespace {
int64_t NowMicros() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// GetTimingStats + PrintStatsAndWriteCSV (lyra_benchmark_lib.cc:62-82,164-182): same statistics, same line format
void PrintStats(const std::vector<int64_t>& t, const char* title) {
  const double n = (double)t.size();
  const double mean = std::accumulate(t.begin(), t.end(), 0.0) / n;
  double var = 0.0;
  for (int64_t v : t) var += ((double)v - mean) * ((double)v - mean) / n;
  std::printf("%18s:  max: %5.3f ms  min: %5.3f ms  mean: %5.3f ms  stdev: %5.3f ms\n", title,
              (double)*std::max_element(t.begin(), t.end()) / 1000.0, (double)*std::min_element(t.begin(), t.end()) / 1000.0,
              mean / 1000.0, std::sqrt(var) / 1000.0);
}
int Bench(const std::string& model_dir, int num_cond_vectors, int num_bits) {
  auto extractor = CreateFeatureExtractor(model_dir);
  auto quantizer = CreateQuantizer(model_dir);
  auto model = CreateGenerativeModel(64, model_dir);
  if (!extractor || !quantizer || !model) { std::fprintf(stderr, "creation failed\n"); return 1; }
  std::mt19937 gen(0x4C797261);
  std::uniform_real_distribution<float> unit(-1.f, 1.f);
  std::vector<int64_t> t_ext, t_q, t_dq, t_gen, t_all;
  std::vector<int16_t> hop(320);
  for (int i = 0; i < num_cond_vectors + 50; ++i) {
    for (auto& s : hop) { float v = unit(gen) * 32768.f; s = (int16_t)std::max(-32768.f, std::min(32767.f, v)); }
    const int64_t t0 = NowMicros();
    auto feats = extractor->Extract(absl::MakeConstSpan(hop.data(), 320));
    const int64_t t1 = NowMicros();
    if (!feats) return 4;
    auto bits = quantizer->Quantize(*feats, num_bits);
    const int64_t t2 = NowMicros();
    if (!bits) return 4;
    auto lossy = quantizer->DecodeToLossyFeatures(*bits);
    const int64_t t3 = NowMicros();
    if (!lossy || !model->AddFeatures(*lossy)) return 4;
    auto pcm = model->GenerateSamples(320);
    const int64_t t4 = NowMicros();
    if (!pcm) return 5;
    if (i < 50) continue;   // warm-up (context creation, first launches)
    t_ext.push_back(t1 - t0); t_q.push_back(t2 - t1); t_dq.push_back(t3 - t2); t_gen.push_back(t4 - t3); t_all.push_back(t4 - t0);
  }
  std::printf("lyra_benchmark at the plugin boundary: 1 stream, %d hops of 20 ms, %d bits; blocking host calls "
              "(H2D + kernels + D2H + synchronise each)\n", num_cond_vectors, num_bits);
  PrintStats(t_ext, "feature_extractor");
  PrintStats(t_q, "quantizer_quantize");
  PrintStats(t_dq, "quantizer_decode");
  PrintStats(t_gen, "model_decode");
  PrintStats(t_all, "total");
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc >= 3 && std::string(argv[1]) == "--bench")
    return Bench(argv[2], argc > 3 ? std::atoi(argv[3]) : 2000, argc > 4 ? std::atoi(argv[4]) : 120);
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
