// plugin_mt_demo.cc -- many codec objects on many threads, each used exactly like the reference uses its plugins (one
// hop per call: lyra_encoder.cc:143-155, lyra_decoder.cc:198-207), served by combined device calls:
//   plugin_mt_demo <model_dir> <pcm_in.s16> <num_streams> <num_bits> <bits_out.txt> <pcm_out.s16>
// pcm_in is [frames][streams][320] int16; thread s owns the extractor / quantizer / generative model of stream s and
// runs all frames of it.  Writes one '0'/'1' line per (frame, stream) and the decoded PCM in the input's layout; prints
// how many plugin calls became how many device calls.
#include <sched.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "lyra_hip_components.h"

using namespace chromemedia::codec;

// A container that is given a CPU *quota* (cgroup v2 cpu.max) smaller than the CPUs it may run on -- 16 CPUs' worth of time on
// a 256-CPU host is what the GPU boxes of this pool have -- spreads a thousand codec threads over all 256 run queues; CFS
// hands the quota out in per-CPU slices, the pool is empty after a few dozen CPUs took one, and every thread that wakes up on
// another CPU waits for the next 100 ms period (cpu.stat: nr_throttled / throttled_usec), whoever it was about to wake up
// included.  Keeping the process on as many CPUs as the quota is worth avoids that.  LYRA_DEMO_CPUS=n overrides, 0 = leave.
static int LimitAffinityToCpuQuota() {
  long want = -1;
  if (const char* e = std::getenv("LYRA_DEMO_CPUS")) want = std::atol(e);
  if (want < 0) {
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota; long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) want = (long)std::ceil(std::atof(quota.c_str()) / (double)period);
  }
  if (want <= 0) return 0;
  cpu_set_t have, keep;
  if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
  if (CPU_COUNT(&have) <= want) return 0;
  CPU_ZERO(&keep);
  long n = 0;
  for (int c = 0; c < CPU_SETSIZE && n < want; ++c) if (CPU_ISSET(c, &have)) { CPU_SET(c, &keep); ++n; }
  return sched_setaffinity(0, sizeof(keep), &keep) == 0 ? (int)n : 0;
}

int main(int argc, char** argv) {
  if (argc != 7) { std::fprintf(stderr, "usage: %s model_dir pcm_in num_streams num_bits bits_out pcm_out\n", argv[0]); return 2; }
  const int cpus = LimitAffinityToCpuQuota();
  if (cpus) std::fprintf(stderr, "affinity limited to %d CPUs (the container's CPU quota)\n", cpus);
  const std::string model_dir = argv[1];
  const int n = std::atoi(argv[3]), num_bits = std::atoi(argv[4]);
  std::ifstream in(argv[2], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  std::vector<int16_t> pcm(raw.size() / 2);
  std::memcpy(pcm.data(), raw.data(), pcm.size() * 2);
  if (n <= 0 || pcm.size() % (size_t)(n * 320) != 0) { std::fprintf(stderr, "bad input size\n"); return 2; }
  const int frames = (int)(pcm.size() / (size_t)(n * 320));
  SetMaxStreams(3 * n);   // three plugin objects (= stream slots) per codec
  std::vector<std::unique_ptr<FeatureExtractorInterface>> ext(n);
  std::vector<std::unique_ptr<VectorQuantizerInterface>> vq(n);
  std::vector<std::unique_ptr<GenerativeModelInterface>> gen(n);
  for (int s = 0; s < n; ++s) {
    ext[s] = CreateFeatureExtractor(model_dir);
    vq[s] = CreateQuantizer(model_dir);
    gen[s] = CreateGenerativeModel(64, model_dir);
    if (!ext[s] || !vq[s] || !gen[s]) { std::fprintf(stderr, "creation failed at stream %d\n", s); return 1; }
  }
  std::vector<std::string> bits((size_t)frames * n);
  std::vector<int16_t> out(pcm.size());
  std::vector<int> rc(n, 0);
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < n; ++s)
    th.emplace_back([&, s] {
      for (int f = 0; f < frames && rc[s] == 0; ++f) {
        const int16_t* hop = pcm.data() + ((size_t)f * n + s) * 320;
        auto feats = ext[s]->Extract(absl::MakeConstSpan(hop, 320));
        if (!feats) { rc[s] = 4; break; }
        auto b = vq[s]->Quantize(*feats, num_bits);
        if (!b) { rc[s] = 4; break; }
        bits[(size_t)f * n + s] = *b;
        auto lossy = vq[s]->DecodeToLossyFeatures(*b);
        if (!lossy || !gen[s]->AddFeatures(*lossy)) { rc[s] = 4; break; }
        auto a = gen[s]->GenerateSamples(320);
        if (!a || a->size() != 320) { rc[s] = 5; break; }
        std::memcpy(out.data() + ((size_t)f * n + s) * 320, a->data(), 640);
      }
    });
  for (auto& t : th) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int s = 0; s < n; ++s)
    if (rc[s]) { std::fprintf(stderr, "stream %d failed (%d)\n", s, rc[s]); return rc[s]; }
  std::ofstream bits_out(argv[5]);
  for (const std::string& b : bits) bits_out << b << "\n";
  std::ofstream pcm_out(argv[6], std::ios::binary);
  pcm_out.write(reinterpret_cast<const char*>(out.data()), out.size() * 2);
  const HipCallStats st = GetHipCallStats();
  std::printf("plugin_calls %ld device_calls %ld largest_batch %ld  (leaders: %.3f s in device calls, %.3f s gathering, %ld gathers timed out)\n",
              st.calls, st.device_calls, st.largest_batch, st.exec_us * 1e-6, st.gather_us * 1e-6, st.gather_timeouts);
  // one hop per plugin call and stream, every call a blocking host call (the reference's plugin contract): what the
  // call-combining layer makes of `n` threads driving `n` codecs
  std::printf("threads %d frames_per_stream %d seconds %.4f frames_per_s %.1f (extract + quantize + dequantize + generate per frame)\n",
              n, frames, secs, (double)n * frames / secs);
  return 0;
}
