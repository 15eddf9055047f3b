// box_probe.hip -- per-instruction-class latency / throughput probe, used to find out what differs between the
// "fast" and "slow" MI355X boxes of the pool (same clocks, yet the int8 stages run 1.4-1.6x slower on some).
// One wave, dependent chains: cycles per op (s_memtime).  Also an 8-wave/SIMD throughput variant per class.
//   hipcc --offload-arch=gfx950 -O2 tools/box_probe.hip -o tools/box_probe.bin && tools/box_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define N 256

__global__ void probe(long long* out, const int* gmem, const int* chain, int seed) {
  __shared__ unsigned char lut[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += blockDim.x) lut[i] = (unsigned char)((i * 7 + 3) & 255);
  __syncthreads();
  long long t0, t1;
  int x = seed + tid;
  // 0. 32-bit integer multiply-add chain (v_mad_u32_u24 / v_mul_lo)
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = x * 1664525 + 1013904223;
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  // 1. mulhi chain
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __mulhi(x, 0x5A17C3D1) + i;
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[1] = t1 - t0;
  // 2. 64-bit mad + shift chain (mbqm_exact shape)
  long long y = x;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) y = ((long long)(int)y * (long long)0x5A17C3D1 + (1ll << 35)) >> ((i & 7) + 30);
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[2] = t1 - t0;
  x ^= (int)y;
  // 3. LDS byte lookup chain
  int idx = x & 1023;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) idx = (lut[idx] * 4 + (idx & 3)) & 1023;
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[3] = t1 - t0;
  x += idx;
  // 4. int8 MFMA dependent chain
  i32x4 a = {x, x + 1, x + 2, x + 3}, b = {x ^ 5, x ^ 6, x ^ 7, x ^ 8}, c = {0, 0, 0, 0};
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[4] = t1 - t0;
  x += c[0] + c[1] + c[2] + c[3];
  // 5. fp32 MFMA dependent chain
  f32x4 cf = {0, 0, 0, 0};
  float af = (float)(x & 7), bf = 0.5f;
  t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, cf, 0, 0, 0);
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[5] = t1 - t0;
  x += (int)cf[0];
  // 6. dependent global load chain (pointer chasing in a 64 KB buffer: L2 hits after the first pass)
  int p = tid & 63;
  for (int i = 0; i < 64; ++i) p = chain[p];
  t0 = clock64();
  for (int i = 0; i < N; ++i) p = chain[p];
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[6] = t1 - t0;
  x += p;
  // 7. dependent scalar load chain (uniform address -> s_load)
  int sp = 0;
  for (int i = 0; i < 16; ++i) sp = __builtin_amdgcn_readfirstlane(gmem[sp]);
  t0 = clock64();
  for (int i = 0; i < N; ++i) sp = __builtin_amdgcn_readfirstlane(gmem[sp]);
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[7] = t1 - t0;
  x += sp;
  // 8. independent int8 MFMAs (throughput, 4 accumulators)
  i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  t0 = clock64();
#pragma unroll 4
  for (int i = 0; i < N / 4; ++i) {
    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
  }
  t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) out[8] = t1 - t0;
  x += c0[0] + c1[1] + c2[2] + c3[3];
  if (x == 0x12345678) out[63] = x;
}

int main() {
  const int CH = 16384;
  std::vector<int> chain(CH), g(CH);
  for (int i = 0; i < CH; ++i) chain[i] = (i * 97 + 31) % CH;   // stride-97 cycle through 64 KB
  for (int i = 0; i < CH; ++i) g[i] = (i * 33 + 17) % CH;
  int *dchain, *dg;
  long long* dout;
  hipMalloc(&dchain, CH * 4); hipMalloc(&dg, CH * 4); hipMalloc(&dout, 64 * 8);
  hipMemcpy(dchain, chain.data(), CH * 4, hipMemcpyHostToDevice);
  hipMemcpy(dg, g.data(), CH * 4, hipMemcpyHostToDevice);
  const char* names[9] = {"mul_lo+add chain", "mul_hi+add chain", "mad64+shift64 chain", "lds u8 lookup chain",
                          "mfma i8 16x16x64 dependent", "mfma f32 16x16x4 dependent", "global load chain (L2)",
                          "scalar load chain", "mfma i8 x4 independent"};
  struct Cfg { int blocks, threads; const char* what; } cfgs[] = {
      {1, 64, "1 wave"}, {256, 64, "1 wave/CU"}, {1024, 512, "8 waves/SIMD-ish (1024 WG x 512 thr)"}};
  for (auto& c : cfgs) {
    long long h[64];
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(dout, 0, 64 * 8);
      hipLaunchKernelGGL(probe, dim3(c.blocks), dim3(c.threads), 0, 0, dout, dg, dchain, rep + 1);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, dout, 64 * 8, hipMemcpyDeviceToHost);
    printf("== %s: cycles per op (WG 0, thread 0)\n", c.what);
    for (int i = 0; i < 9; ++i) printf("   %-32s %7.1f\n", names[i], (double)h[i] / N);
  }
  return 0;
}
