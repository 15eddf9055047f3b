// mfma_valu_overlap_probe.hip -- does VALU work hide under fp32 / int8 MFMAs on gfx950, within a wave and across the waves of a
// SIMD?  One workgroup of 256 * W threads per CU (W waves per SIMD).  Each wave runs REP groups of
//     M matrix instructions (independent accumulators, round robin) + V vector instructions (independent),
// interleaved one by one; or, in the "split" mode, the even waves of a SIMD run only the matrix part and the odd waves only the
// vector part.  Reported: shader cycles per group seen by wave 0 and SIMD cycles per group from wall time.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_overlap_probe.hip -o tools/mfma_valu_overlap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ long long g_cyc;

template <int KIND, int V, int vkind, int split>   // KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_i32_16x16x64_i8;  V vector instructions per matrix instruction
__global__ __launch_bounds__(1024) void probe(float* out, int rep) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  i4 iacc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  i4 ia = {(int)threadIdx.x, 2, 3, 4}, ib = {5, 6, 7, 8};
  float v[8]; int w[8];
  for (int i = 0; i < 8; ++i) { v[i] = a + i; w[i] = threadIdx.x + i; }
  const int wave = threadIdx.x >> 6;
  const bool do_m = !split || ((wave >> 2) & 1) == 0;   // waves 0-3 sit on SIMDs 0-3, waves 4-7 on SIMDs 0-3 again, ...
  const bool do_v = !split || ((wave >> 2) & 1) == 1;
  long long t0 = clock64();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (do_m) {
        if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(iacc[g]) : "v"(ia), "v"(ib));
      }
      if (do_v) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          if (vkind == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(g * V + k) & 7]) : "v"(b));
          else asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(w[(g * V + k) & 7]) : "v"(w[7 - ((g * V + k) & 7)]));
        }
      }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) g_cyc = t1 - t0;
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i] + (float)w[i];
  for (int g = 0; g < 4; ++g) s += acc[g][0] + (float)iacc[g][0];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int KIND, int V, int vkind, int split>
static void run1(const char* name, float* out) {
  const int rep = 4096;
  hipEvent_t ea, eb;
  hipEventCreate(&ea); hipEventCreate(&eb);
    {
      printf("%-28s V=%d per matrix op, vector = %-14s %-38s", name, V, vkind ? "v_mad_i32_i24" : "v_add_f32",
             split ? "matrix waves + vector waves (split)" : "interleaved in every wave");
      for (int wps : {1, 2, 4}) {
        if (split && wps == 1) { printf(" %20s", "-"); continue; }
        hipLaunchKernelGGL((probe<KIND, V, vkind, split>), dim3(256), dim3(256 * wps), 0, 0, out, 16);
        hipDeviceSynchronize();
        hipEventRecord(ea, 0);
        hipLaunchKernelGGL((probe<KIND, V, vkind, split>), dim3(256), dim3(256 * wps), 0, 0, out, rep);
        hipEventRecord(eb, 0);
        hipEventSynchronize(eb);
        float ms = 0;
        hipEventElapsedTime(&ms, ea, eb);
        long long cyc = 0;
        hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cyc), 8);
        // groups executed per SIMD: interleaved: wps waves x rep x 4; split: (wps / 2) matrix waves and (wps / 2) vector waves
        const double groups = (double)rep * 4 * (split ? wps / 2 : wps);
        printf("   %dw: %6.1f | %6.1f", wps, (double)cyc / (rep * 4.0), ms * 1e-3 * 2.4e9 / groups);
      }
      printf("\n");
    }
}

template <int KIND, int V>
static void run(const char* name, float* out) {
  run1<KIND, V, 0, 0>(name, out); run1<KIND, V, 0, 1>(name, out);
  if (V) { run1<KIND, V, 1, 0>(name, out); run1<KIND, V, 1, 1>(name, out); }
}

int main() {
  float* out;
  hipMalloc(&out, 8192);
  printf("per cell: shader cycles per (1 matrix op + V vector ops) seen by wave 0 | SIMD cycles per group from wall time at 2.4 GHz\n"
         "(split: per pair of one matrix-wave group and one vector-wave group)\n");
  run<0, 0>("v_mfma_f32_16x16x4_f32", out);
  run<0, 2>("v_mfma_f32_16x16x4_f32", out);
  run<0, 4>("v_mfma_f32_16x16x4_f32", out);
  run<0, 8>("v_mfma_f32_16x16x4_f32", out);
  run<1, 0>("v_mfma_i32_16x16x64_i8", out);
  run<1, 4>("v_mfma_i32_16x16x64_i8", out);
  run<1, 8>("v_mfma_i32_16x16x64_i8", out);
  return 0;
}
