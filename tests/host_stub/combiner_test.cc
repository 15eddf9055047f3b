// combiner_test.cc -- TEST: the C++ plugin layer's host logic on the CPU, against fake_lyra_hip.cc.
//   combiner_test <threads> <hops>
// Every thread owns an extractor, a quantizer, a generative model and a log-mel extractor and uses them hop by hop with
// its own bit rate.  Checks: every result equals what the fake ABI gives for that stream alone; no two device calls
// overlapped; calls were combined (fewer device calls than plugin calls) yet every plugin call was served exactly once.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "lyra_hip_components.h"

extern "C" long fake_device_calls();
extern "C" long fake_device_rows();
extern "C" int fake_overlapping_calls();
using namespace chromemedia::codec;

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 24, hops = argc > 2 ? std::atoi(argv[2]) : 20;
  SetMaxStreams(4 * n);
  {   // contexts are created per side, on demand (round 6): an encode-only or decode-only process holds ONE
    if (GetHipContextCount() != 0) return 20;
    {
      auto ext = CreateFeatureExtractor("unused");
      auto vq = CreateQuantizer("unused");                       // either side will do: the extractor side exists
      if (!ext || !vq || GetHipContextCount() != 1) return 21;
      std::vector<int16_t> pcm(320, 5);
      auto f = ext->Extract(absl::MakeConstSpan(pcm.data(), 320));
      if (!f || !vq->Quantize(*f, 64) || GetHipContextCount() != 1) return 22;
      if (!vq->DecodeToLossyFeatures(std::string(64, '0')) || GetHipContextCount() != 2) return 23;   // first decoder-side call
    }
    if (GetHipContextCount() != 0) return 24;                    // last object gone: everything released
    {
      auto gen = CreateGenerativeModel(64, "unused");
      auto vq = CreateQuantizer("unused");                       // a decoder: generative model first (lyra_decoder.cc:117-138)
      if (!gen || !vq || GetHipContextCount() != 1) return 25;
      auto lossy = vq->DecodeToLossyFeatures(std::string(120, '1'));
      if (!lossy || !gen->AddFeatures(*lossy) || !gen->GenerateSamples(320) || GetHipContextCount() != 1) return 26;
    }
    if (GetHipContextCount() != 0) return 27;
  }
  const HipCallStats before = GetHipCallStats();
  const long rows_before = fake_device_rows();
  std::vector<int> bad(n, 0);
  std::vector<std::thread> th;
  for (int s = 0; s < n; ++s)
    th.emplace_back([&, s] {
      auto ext = CreateFeatureExtractor("unused");
      auto vq = CreateQuantizer("unused");
      auto gen = CreateGenerativeModel(64, "unused");
      auto mel = CreateLogMelExtractor("unused");
      if (!ext || !vq || !gen || !mel) { bad[s] = 1; return; }
      const int num_bits = s % 3 == 0 ? 64 : (s % 3 == 1 ? 120 : 184);
      float slot_base = -1.f;
      for (int h = 0; h < hops && !bad[s]; ++h) {
        std::vector<int16_t> pcm(320);
        for (int i = 0; i < 320; ++i) pcm[i] = (int16_t)(s * 7 + h * 13 + i);
        auto f = ext->Extract(absl::MakeConstSpan(pcm.data(), 320));
        if (!f || f->size() != 64) { bad[s] = 2; break; }
        // the fake adds 1000 * (the extractor object's stream slot), which the plugin interface hides: it must be the
        // same whole multiple of 1000 for every element and every hop of this object
        const float base = (*f)[0] - (float)pcm[0];
        if (base < 0.f || base != 1000.f * (float)(int)(base / 1000.f) || (slot_base >= 0.f && base != slot_base)) bad[s] = 3;
        slot_base = base;
        for (int i = 0; i < 64; ++i)
          if ((*f)[i] != base + (float)pcm[i] + (float)i * 0.5f) bad[s] = 3;
        auto bits = vq->Quantize(*f, num_bits);
        if (!bits || (int)bits->size() != num_bits) { bad[s] = 4; break; }
        for (int k = 0; k < num_bits / 4; ++k) {
          const int want = (((int)(*f)[k] + k) & 15);
          int got = 0;
          for (int b = 0; b < 4; ++b) got = got * 2 + ((*bits)[4 * k + b] == '1');
          if (got != want) bad[s] = 5;
        }
        auto lossy = vq->DecodeToLossyFeatures(*bits);
        if (!lossy || lossy->size() != 64) { bad[s] = 6; break; }
        if (!gen->AddFeatures(*lossy)) { bad[s] = 7; break; }
        auto a = gen->GenerateSamples(320);
        if (!a || a->size() != 320) { bad[s] = 8; break; }
        auto m = mel->Extract(absl::MakeConstSpan(pcm.data(), 320));
        if (!m || m->size() != 160) { bad[s] = 9; break; }
        if (ext->Extract(absl::MakeConstSpan(pcm.data(), 319)).has_value()) bad[s] = 10;   // validation stays per call
        if (vq->Quantize(*f, 62).has_value()) bad[s] = 11;
      }
    });
  for (auto& t : th) t.join();
  for (int s = 0; s < n; ++s)
    if (bad[s]) { std::fprintf(stderr, "thread %d failed check %d\n", s, bad[s]); return 1; }
  HipCallStats st = GetHipCallStats();
  st.calls -= before.calls;
  st.device_calls -= before.device_calls;
  std::printf("plugin_calls %ld device_calls %ld largest_batch %ld fake_device_calls %ld fake_rows %ld overlapping %d\n",
              st.calls, st.device_calls, st.largest_batch, fake_device_calls(), fake_device_rows() - rows_before, fake_overlapping_calls());
  if (fake_overlapping_calls() != 0) return 2;
  if (st.calls != 5L * n * hops || fake_device_rows() - rows_before != st.calls) return 3;
  return 0;
}
