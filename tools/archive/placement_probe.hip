// placement_probe.hip -- where do the workgroups of SMALL concurrent dispatches land?
// Config #2 (1,024 streams) runs kernels of 64-256 workgroups on a 256-CU chip, two (or more) at a time on different
// HIP streams.  This probe launches spin kernels shaped like the stage kernels (threads, LDS bytes, workgroup count)
// on 1 / 2 / 7 streams and records, per workgroup, XCC / SE / CU from the hardware id registers and the start / end
// wall clock; the report says how many distinct CUs each dispatch used, how many workgroups shared a CU across
// dispatches, and how long each dispatch took.  Also: streams created with complementary CU masks.
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };

__global__ void spin(Rec* out, int spin_ticks) {   // spin_ticks: 100 MHz wall-clock ticks
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  float x = (float)threadIdx.x;
  lds[threadIdx.x] = x;
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f;
  }
  lds[threadIdx.x] += x;
  if (threadIdx.x == 0) { out[blockIdx.x] = Rec{hw, xcc, t0, (unsigned long long)wall_clock64()}; }
}

struct Shape { const char* name; int wgs, threads, lds; };

static int cu_key(const Rec& r) {
  const unsigned cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 15;
  return (int)(((xcc * 8 + se) * 2 + sh) * 16 + cu);
}

static void run(const char* title, const std::vector<Shape>& shapes, const std::vector<hipStream_t>& streams, int spin_ticks) {
  const int n = (int)shapes.size();
  std::vector<Rec*> d(n);
  for (int i = 0; i < n; ++i) CHECK(hipMalloc(&d[i], sizeof(Rec) * shapes[i].wgs));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep)   // second repetition is the one reported (first warms code / clocks)
    for (int i = 0; i < n; ++i)
      hipLaunchKernelGGL(spin, dim3(shapes[i].wgs), dim3(shapes[i].threads), shapes[i].lds, streams[i], d[i], spin_ticks);
  CHECK(hipDeviceSynchronize());
  printf("== %s\n", title);
  std::map<int, int> per_cu_all;
  std::vector<std::vector<Rec>> h(n);
  unsigned long long tmin = ~0ull;
  for (int i = 0; i < n; ++i) {
    h[i].resize(shapes[i].wgs);
    CHECK(hipMemcpy(h[i].data(), d[i], sizeof(Rec) * shapes[i].wgs, hipMemcpyDeviceToHost));
    for (auto& r : h[i]) if (r.t0 < tmin) tmin = r.t0;
  }
  for (int i = 0; i < n; ++i) {
    std::map<int, int> per_cu;
    std::set<unsigned> xccs;
    unsigned long long a = ~0ull, b = 0;
    for (auto& r : h[i]) { per_cu[cu_key(r)]++; per_cu_all[cu_key(r)]++; xccs.insert(r.xcc & 15); if (r.t0 < a) a = r.t0; if (r.t1 > b) b = r.t1; }
    int mx = 0;
    for (auto& kv : per_cu) if (kv.second > mx) mx = kv.second;
    printf("  %-10s %4d WGs x %3d thr %6d B LDS: %3zu CUs on %zu XCCs, max %d WGs/CU, start %+7.1f us, span %6.1f us\n", shapes[i].name,
           shapes[i].wgs, shapes[i].threads, shapes[i].lds, per_cu.size(), xccs.size(), mx, (double)(a - tmin) / 100.0, (double)(b - a) / 100.0);
  }
  int hist[16] = {0};
  for (auto& kv : per_cu_all) hist[kv.second < 15 ? kv.second : 15]++;
  printf("  all dispatches together: %zu distinct CUs;  CUs holding k workgroups:", per_cu_all.size());
  for (int k = 1; k < 16; ++k) if (hist[k]) printf("  k=%d: %d", k, hist[k]);
  printf("\n");
  for (int i = 0; i < n; ++i) CHECK(hipFree(d[i]));
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int lo, hi;
  CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::vector<hipStream_t> s(8), shi(8);
  for (auto& x : s) CHECK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, lo));
  for (auto& x : shi) CHECK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, hi));
  const int T = 4000;   // 40 us
  const Shape e0{"enc_s0", 256, 256, 29 * 1024}, e1{"enc_s1", 128, 512, 35 * 1024}, e2{"enc_s2", 128, 512, 54 * 1024},
      rq{"rvq_enc", 64, 256, 26 * 1024}, d0{"dec_s0", 128, 512, 68 * 1024}, d1{"dec_s1", 128, 512, 46 * 1024}, d2{"dec_s2", 256, 256, 32 * 1024};
  run("one dispatch of 128 WGs", {e1}, {s[0]}, T);
  run("one dispatch of 64 WGs", {rq}, {s[0]}, T);
  run("one dispatch of 256 WGs", {e0}, {s[0]}, T);
  run("two streams, equal priority: enc_s1 | dec_s1", {e1, d1}, {s[0], s[1]}, T);
  run("two streams, second at high priority: enc_s1 | dec_s1", {e1, d1}, {s[0], shi[1]}, T);
  run("two streams: enc_s0 | dec_s0", {e0, d0}, {s[0], shi[1]}, T);
  run("three streams: enc_s1 | rvq | dec_s1", {e1, rq, d1}, {s[0], s[2], shi[1]}, T);
  run("seven streams, one stage kernel each (decoder stages high priority)", {e0, e1, e2, rq, d0, d1, d2},
      {s[0], s[1], s[2], s[3], shi[4], shi[5], shi[6]}, T);
  run("seven streams, equal priority", {e0, e1, e2, rq, d0, d1, d2}, {s[0], s[1], s[2], s[3], s[4], s[5], s[6]}, T);
  run("four streams x 64 WGs (sub-batches of 512 streams)", {{"a", 64, 512, 35 * 1024}, {"b", 64, 512, 35 * 1024}, {"c", 64, 512, 46 * 1024}, {"d", 64, 512, 46 * 1024}},
      {s[0], s[1], shi[2], shi[3]}, T);
  if (!getenv("NO_CU_MASK")) {
    // complementary CU masks, period 16 bits (8 on / 8 off): an even share of every XCC whether the runtime numbers the mask
    // bits XCC-major or interleaves them across XCCs
    const int words = (p.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> ma(words, 0x00ff00ffu), mb(words, 0xff00ff00u);
    hipStream_t ca, cb;
    CHECK(hipExtStreamCreateWithCUMask(&ca, words, ma.data()));
    CHECK(hipExtStreamCreateWithCUMask(&cb, words, mb.data()));
    run("CU-masked streams (0x00ff00ff | 0xff00ff00): enc_s1 | dec_s1", {e1, d1}, {ca, cb}, T);
    run("CU-masked streams: enc_s0 | dec_s2", {e0, d2}, {ca, cb}, T);
  }
  return 0;
}
