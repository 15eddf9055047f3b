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

// ---- throughput under full occupancy: every thread runs 4 independent chains of one instruction class ----------
template <int CLS>
__global__ void tput(int* out, int seed, int iters) {
  __shared__ unsigned char lut[4096];
  const int tid = threadIdx.x;
  if (CLS == 4 || CLS == 5) {
    for (int i = tid; i < 4096; i += blockDim.x) lut[i] = (unsigned char)((i * 7 + 3) & 255);
    __syncthreads();
  }
  int a = seed + tid, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3;
  long long la = a, lb = b, lc = c, ld = d;
  i32x4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
  f32x4 f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0;
  const i32x4 av = {a, b, c, d}, bv = {d, c, b, a};
  for (int i = 0; i < iters; ++i) {
    if (CLS == 0) { a = a * 3 + i; b = b * 5 + i; c = c * 7 + i; d = d * 9 + i; a ^= a >> 3; b ^= b >> 5; c ^= c >> 7; d ^= d >> 9; }  // plain 32-bit VALU mix
    if (CLS == 1) { a = __mulhi(a, 0x5A17C3D1) + i; b = __mulhi(b, 0x6B28D4E3) + i; c = __mulhi(c, 0x7C39E5F5) + i; d = __mulhi(d, 0x4D4AF607) + i; }
    if (CLS == 2) {
      la = ((long long)(int)la * 0x5A17C3D1ll + (1ll << 35)) >> ((i & 7) + 30);
      lb = ((long long)(int)lb * 0x6B28D4E3ll + (1ll << 35)) >> ((i & 7) + 30);
      lc = ((long long)(int)lc * 0x7C39E5F5ll + (1ll << 35)) >> ((i & 7) + 30);
      ld = ((long long)(int)ld * 0x4D4AF607ll + (1ll << 35)) >> ((i & 7) + 30);
    }
    if (CLS == 3) { a = (a << 8 >> 24) * (b << 16 >> 24) + c; b = (b << 8 >> 24) * (c << 16 >> 24) + d; c = (c << 8 >> 24) * (d << 16 >> 24) + a; d = (d << 8 >> 24) * (a << 16 >> 24) + b; }  // bfe + mad
    if (CLS == 4) { a = lut[a & 4095] + (a >> 1) + i; b = lut[b & 4095] + (b >> 1) + i; c = lut[c & 4095] + (c >> 1) + i; d = lut[d & 4095] + (d >> 1) + i; }
    if (CLS == 5) {
      i32x4 v = *reinterpret_cast<const i32x4*>(&lut[(a & 255) * 16]);
      a += v[0] + v[3] + i; b += v[1]; c += v[2]; d += v[3];
    }
    if (CLS == 6) {
      m0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, m1, 0, 0, 0);
      m2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, m3, 0, 0, 0);
    }
    if (CLS == 7) {
      f0 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, f0, 0, 0, 0); f1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, f2, 0, 0, 0); f3 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, f3, 0, 0, 0);
    }
  }
  int r = a + b + c + d + (int)(la + lb + lc + ld) + m0[0] + m1[1] + m2[2] + m3[3] + (int)(f0[0] + f1[1] + f2[2] + f3[3]);
  if (r == 0x12345678) out[0] = r;
}

// ---- instruction fetch: the same 8192 VALU ops per pass as straight-line code (~64 KB, larger than a CU pair's
// instruction cache) versus a compact loop
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
__global__ void bigcode(int* out, int seed, int passes) {
  int a = seed + threadIdx.x, b = a * 3 + 1;
  for (int p = 0; p < passes; ++p) {
    R256(R16(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0" : "+v"(a), "+v"(b));))
  }
  if (a + b == 0x12345678) out[0] = a;
}
__global__ void smallcode(int* out, int seed, int passes) {
  int a = seed + threadIdx.x, b = a * 3 + 1;
  for (int p = 0; p < passes * 256; ++p) {
    R16(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %1, %1, %0" : "+v"(a), "+v"(b));)
  }
  if (a + b == 0x12345678) out[0] = a;
}
template <class K>
static void run_code(const char* name, K kern, int* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, dout, 1, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(512), dim3(512), 0, 0, dout, 2, 20);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("   %-36s %8.3f ms\n", name, ms);
}

// ---- scattered HBM reads (TLB reach / fragment size): every thread reads 16 bytes from `touches` pseudo-random
// 128-byte lines of a buffer of `span_mb` MB
__global__ void scatter(const int* buf, unsigned lines, int touches, int* out) {
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  int acc = 0;
  for (int i = 0; i < touches; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned l = (x >> 4) % lines;
    acc += buf[(size_t)l * 32 + (threadIdx.x & 3)];
  }
  if (acc == 0x12345678) out[0] = acc;
}
static void run_scatter(const int* buf, size_t span_mb, int* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned lines = (unsigned)(span_mb * 1024 * 1024 / 128);
  const int blocks = 2048, threads = 256, touches = 16;
  hipLaunchKernelGGL(scatter, dim3(blocks), dim3(threads), 0, 0, buf, lines, touches, dout);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(scatter, dim3(blocks), dim3(threads), 0, 0, buf, lines, touches, dout);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("   random 128 B lines over %4zu MB: %8.3f ms  (%.1f G lines/s)\n", span_mb, ms,
         (double)blocks * threads * touches / ms / 1e6);
}

template <int CLS>
static void run_tput(const char* name, int* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2048, blocks = 2048, threads = 256;
  hipLaunchKernelGGL(tput<CLS>, dim3(blocks), dim3(threads), 0, 0, dout, 1, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(tput<CLS>, dim3(blocks), dim3(threads), 0, 0, dout, 2, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // per-SIMD cycles per wave-iteration at a nominal 2.4 GHz: waves = blocks*threads/64 over 1024 SIMDs
  double waves_per_simd = (double)blocks * threads / 64 / 1024.0;
  printf("   %-36s %8.3f ms   %7.1f cycles/iter/wave-slot @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (iters * waves_per_simd));
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
  printf("== throughput, 2048 WG x 256 threads, 4 independent chains per thread\n");
  int* dres;
  hipMalloc(&dres, 64);
  run_tput<0>("32-bit VALU mix (8 ops/iter)", dres);
  run_tput<1>("mul_hi + add x4", dres);
  run_tput<2>("mad64 + shift64 x4", dres);
  run_tput<3>("bfe + mad x4", dres);
  run_tput<4>("lds u8 lookup x4", dres);
  run_tput<5>("lds b128 read", dres);
  run_tput<6>("mfma i8 16x16x64 x4", dres);
  run_tput<7>("mfma f32 16x16x4 x4", dres);
  printf("== scattered reads\n");
  {
    int* big;
    hipMalloc(&big, (size_t)1024 << 20);
    hipMemset(big, 1, (size_t)1024 << 20);
    run_scatter(big, 2, dres);
    run_scatter(big, 32, dres);
    run_scatter(big, 256, dres);
    run_scatter(big, 1024, dres);
    hipFree(big);
  }
  printf("== instruction fetch: 512 WG x 512 threads, 20 passes of 8192 VALU ops\n");
  run_code("straight-line 64 KB body", bigcode, dres);
  run_code("compact loop", smallcode, dres);
  return 0;
}
