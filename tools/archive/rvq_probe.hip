// rvq_probe.hip -- where does rvq_encode's time go?  A copy of the product kernel (lyra_amd/csrc/misc_kernels.hip)
// with parts switched off by a template mask, timed with HIP events at B = 4096, plus the workgroup -> CU placement.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/rvq_probe.hip -o tools/rvq_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LYRA_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const T LYRA_GLOBAL* as_global(const T* p) { return (const T LYRA_GLOBAL*)p; }
__device__ unsigned g_hwid[4096];

// ABL: ablation mask for tools/rvq_probe.hip (0 in the product): 1 no cross-stage row prefetch (rows read in the chain),
// 2 no argmin (winner = lane-invariant function of the sum), 4 no residual update / write-back, 8 residual kept in
// registers (no LDS broadcast reads; wrong results), 16 no codebook window traffic after the first window.
template <int ABL, int FPW, int W = 4>
__device__ __forceinline__ void rvq_encode_body(const float* __restrict__ cb, const float* __restrict__ feats, int B,
                                                int num_stages, int32_t* __restrict__ indices,
                                                uint8_t* __restrict__ packets) {
  constexpr int ROW = 68, WFLOATS = W * 16 * ROW;
  __shared__ __attribute__((aligned(16))) float cbs[3][WFLOATS];
  __shared__ __attribute__((aligned(16))) float rs[16][ROW];
  const int tid = threadIdx.x;
  if (tid == 0) g_hwid[blockIdx.x] = (__builtin_amdgcn_s_getreg(63492) & 0xffff) | (__builtin_amdgcn_s_getreg(63508) << 16);
  constexpr int NT = 1024 / FPW, CH = W * 256 / NT;   // threads per workgroup; f32x4 chunks per thread per stage window
  const int lane = tid & 63, wave = tid >> 6;
  const bool active = lane < 16 * FPW;
  const int j = lane & 15;
  const int fslot = wave * FPW + ((lane >> 4) & (FPW - 1));
  const int frame = blockIdx.x * 16 + fslot;
  const int f = min(frame, B - 1);
  // staging: the window's W x [16][64] floats = W * 256 f32x4 chunks, CH per thread
  const f32x4 LYRA_GLOBAL* cbg = reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(cb));
  f32x4 stage_in[CH];
  auto gload = [&](int win) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int chunk = tid + i * NT, v = chunk >> 8;
      stage_in[i] = cbg[(size_t)min(win * W + v, 45) * 256 + (chunk & 255)];
    }
  };
  auto lstore = [&](int win) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int chunk = tid + i * NT, v = chunk >> 8, row = (chunk >> 4) & 15, c4 = chunk & 15;
      *reinterpret_cast<f32x4*>(cbs[win % 3] + v * 16 * ROW + row * ROW + c4 * 4) = stage_in[i];
    }
  };
  gload(0);
  float* rme = rs[fslot];
  f32x4 mine = *reinterpret_cast<const f32x4*>(&feats[(size_t)f * 64 + j * 4]);   // this lane's four residual dims
  if (active) *reinterpret_cast<f32x4*>(&rme[j * 4]) = mine;
  lstore(0);
  gload(1);
  __syncthreads();
  const int nbytes = (num_stages + 1) >> 1;
  int cur = 0;
  f32x4 rowa[16], rowb[16];
  auto load_row = [&](f32x4 (&row)[16], int k) {
    const float* c = cbs[(k / W) % 3] + (k & (W - 1)) * 16 * ROW + j * ROW;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) row[d4] = *reinterpret_cast<const f32x4*>(&c[d4 * 4]);
  };
  if (active) load_row(rowa, 0);
  auto stage = [&](int k, const f32x4 (&row)[16], f32x4 (&next)[16]) {
    const int u = k & (W - 1), win = k / W;
    if (u == 0 && (!(ABL & 16) || win == 0)) {
      // window win+1 -> LDS (fetched a window ago), request window win+2.  Buffer (win+1) % 3 last held window
      // win-2, which nobody reads any more: every wave passed the previous window's barrier, i.e. finished
      // window win-2, before any wave could get here.
      lstore(win + 1);
      gload(win + 2);
      __syncthreads();
    }
    if (!active) return;
    asm volatile("" ::: "memory");   // rs is rewritten by the other lanes of the frame: never carry it in registers
    float sum = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
      const f32x4 rv = (ABL & 8) ? mine * (float)(d4 + 1) : *reinterpret_cast<const f32x4*>(&rme[d4 * 4]);
      const f32x4 cv = (ABL & 1) ? *reinterpret_cast<const f32x4*>(&(cbs[win % 3] + u * 16 * ROW + j * ROW)[d4 * 4]) : row[d4];
      const f32x4 df = rv - cv;
      const f32x4 sq = df * df;
      sum = sum + sq[0];
      sum = sum + sq[1];
      sum = sum + sq[2];
      sum = sum + sq[3];
    }
    // Off the critical path: issued after the chain (the residual reads above must not queue behind it) and before
    // the reduction, so the 16 reads drain while the DPP steps run and the winner-row read below finds the LDS idle.
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 1) && k + 1 < num_stages) load_row(next, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ARG_MIN = first minimum.  Branch-free 16-lane all-reduce with DPP row rotations (register-only, no LDS
    // crossbar): the row minimum of the distance, then the lowest lane index among the lanes that hold it.
    // Distances are sums of squares (>= 0, finite for finite features); fminf returns one of its operands exactly.
    float m = sum;
#define LYRA_ROR_MINF(N) \
    m = __builtin_fminf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x120 + (N), 0xf, 0xf, false)));
    if (!(ABL & 2)) { LYRA_ROR_MINF(8) LYRA_ROR_MINF(4) LYRA_ROR_MINF(2) LYRA_ROR_MINF(1) }
#undef LYRA_ROR_MINF
    int best = 0;
    if (ABL & 32) {   // lowest lane of the frame's 16-lane field that holds the minimum: ballot + find-first-set
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(sum == m);
      best = __builtin_ctz((unsigned)(bal >> (lane & 48)) | 0x10000u);
    }
    if (!(ABL & 2) && !(ABL & 32)) best = sum == m ? j : 16;
    if (!(ABL & 2) && !(ABL & 32)) {
#define LYRA_ROR_MINI(N) best = min(best, __builtin_amdgcn_update_dpp(0, best, 0x120 + (N), 0xf, 0xf, false));
    LYRA_ROR_MINI(8) LYRA_ROR_MINI(4) LYRA_ROR_MINI(2) LYRA_ROR_MINI(1)
#undef LYRA_ROR_MINI
    best &= 15;   // (only reachable with NaN distances: keep the LDS address in range)
    }
    if (ABL & 2) best = (__builtin_bit_cast(int, sum) >> 3) & 15;
    if (!(ABL & 4)) {  // r <- r - (r + (q - r))
      const float* c = cbs[win % 3] + u * 16 * ROW;
      const f32x4 qv = *reinterpret_cast<const f32x4*>(&c[best * ROW + j * 4]);
      const f32x4 t1 = qv - mine;
      const f32x4 t2 = mine + t1;
      mine = mine - t2;
      *reinterpret_cast<f32x4*>(&rme[j * 4]) = mine;
    }
    if (j == 0 && frame < B) {
      if (indices) indices[(size_t)frame * 46 + k] = best;
      if (packets) {
        if (k & 1) packets[(size_t)frame * nbytes + (k >> 1)] = (uint8_t)(cur | best);
        else cur = best << 4;
      }
    }
  };
#pragma unroll 1
  for (int k = 0; k < num_stages; k += 2) {
    stage(k, rowa, rowb);
    if (k + 1 < num_stages) stage(k + 1, rowb, rowa);
  }
  if (j == 0 && frame < B) {
    if (packets && (num_stages & 1)) packets[(size_t)frame * nbytes + (num_stages >> 1)] = (uint8_t)cur;
    if (indices)
      for (int k = num_stages; k < 46; ++k) indices[(size_t)frame * 46 + k] = -1;
  }
}


template <int ABL, int FPW, int W>
__global__ __launch_bounds__(1024 / FPW) void k(const float* cb, const float* feats, int B, int ns, int32_t* idx, uint8_t* pk) {
  rvq_encode_body<ABL, FPW, W>(cb, feats, B, ns, idx, pk);
}
template <int ABL, int FPW = 4, int W = 4>
void run(const char* what, const float* cb, const float* feats, int B, int32_t* idx, uint8_t* pk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<ABL, FPW, W>), dim3((B + 15) / 16), dim3(1024 / FPW), 0, 0, cb, feats, B, 46, idx, pk);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ABL, FPW, W>), dim3((B + 15) / 16), dim3(1024 / FPW), 0, 0, cb, feats, B, 46, idx, pk);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-58s %7.1f us\n", what, ms / 20 * 1e3);
}
int main(int argc, char** argv) {
  int B = argc > 1 ? atoi(argv[1]) : 4096;
  std::vector<float> hcb(46 * 16 * 64), hf((size_t)B * 64);
  srand(1);
  for (auto& v : hcb) v = (rand() / (float)RAND_MAX - 0.5f) * 4;
  for (auto& v : hf) v = (rand() / (float)RAND_MAX - 0.5f) * 40;
  float *cb, *feats; int32_t* idx; uint8_t* pk;
  hipMalloc(&cb, hcb.size() * 4); hipMalloc(&feats, hf.size() * 4); hipMalloc(&idx, (size_t)B * 46 * 4); hipMalloc(&pk, (size_t)B * 23);
  hipMemcpy(cb, hcb.data(), hcb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(feats, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
  run<0>("full", cb, feats, B, nullptr, pk);
  unsigned h[4096]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hwid), sizeof h);
  std::map<unsigned, int> cnt;
  for (int i = 0; i < (B + 15) / 16; ++i) cnt[((h[i] >> 16) & 15) << 16 | ((h[i] >> 8) & 0xff)]++;
  std::map<int, int> hist; for (auto& kv : cnt) hist[kv.second]++;
  printf("workgroups %d on %zu CUs; WGs/CU histogram:", (B + 15) / 16, cnt.size()); for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first); printf("\n");
  run<0, 4, 8>("full, W = 8", cb, feats, B, nullptr, pk);
  run<32, 4, 4>("full, ballot argmin", cb, feats, B, nullptr, pk);
  run<32, 4, 8>("full, ballot argmin, W = 8", cb, feats, B, nullptr, pk);
  run<0, 2>("full, 2 frames per wave (512 threads)", cb, feats, B, nullptr, pk);
  run<0, 1>("full, 1 frame per wave (1024 threads)", cb, feats, B, nullptr, pk);
  run<2 | 4 | 8 | 16, 2>("chain only, 2 frames per wave", cb, feats, B, nullptr, pk);
  run<2 | 4 | 8 | 16, 1>("chain only, 1 frame per wave", cb, feats, B, nullptr, pk);
  run<1, 2>("no cross-stage row prefetch, 2 frames per wave", cb, feats, B, nullptr, pk);
  run<1>("no cross-stage row prefetch", cb, feats, B, nullptr, pk);
  run<2>("no argmin", cb, feats, B, nullptr, pk);
  run<4>("no update / write-back", cb, feats, B, nullptr, pk);
  run<8>("no residual broadcast reads", cb, feats, B, nullptr, pk);
  run<16>("no codebook window traffic (one window)", cb, feats, B, nullptr, pk);
  run<2 | 4>("no argmin, no update", cb, feats, B, nullptr, pk);
  run<2 | 4 | 8>("no argmin, no update, no residual reads", cb, feats, B, nullptr, pk);
  run<2 | 4 | 8 | 16>("chain only (rows prefetched from one window)", cb, feats, B, nullptr, pk);
  run<1 | 2 | 4 | 8 | 16>("chain only, rows read in the chain", cb, feats, B, nullptr, pk);
  std::vector<uint8_t> a((size_t)B * 23), b2((size_t)B * 23);
  hipLaunchKernelGGL((k<0, 4, 4>), dim3((B + 15) / 16), dim3(256), 0, 0, cb, feats, B, 46, idx, pk); hipDeviceSynchronize();
  hipMemcpy(a.data(), pk, a.size(), hipMemcpyDeviceToHost);
  hipLaunchKernelGGL((k<32, 4, 8>), dim3((B + 15) / 16), dim3(256), 0, 0, cb, feats, B, 46, idx, pk); hipDeviceSynchronize();
  hipMemcpy(b2.data(), pk, b2.size(), hipMemcpyDeviceToHost);
  printf("ballot/W=8 variant packets %s\n", a == b2 ? "identical" : "DIFFER");
  return 0;
}
