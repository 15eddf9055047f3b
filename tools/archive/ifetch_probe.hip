// ifetch_probe.hip -- what does ONCE-EXECUTED straight-line code cost on gfx950?  The stage kernels are 8-35 KB of code
// that every wave executes exactly once; an instruction-cache line that no wave of the CU has touched yet has to come from
// L2 (or further).  Three kernels issue the same N dependent-free VALU instructions per wave:
//   straight : N distinct instructions in a row (N * 8 bytes of code: v_add_f32 with a literal = 8-byte encoding)
//   looped   : the same N instructions as N / 256 trips over a 256-instruction body
//   straight, second pass inside one launch : the code is in the instruction cache already
// 512 workgroups x 512 threads (2 per CU, 16 waves per CU), as dec_s0 / enc_s2 run.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
#define REP4096(x) REP16(REP256(x))
// 8-byte VOP2 with a 32-bit literal; the four accumulators keep consecutive instructions independent
#define I4 "v_add_f32 %0, 0x3f800001, %0\n v_add_f32 %1, 0x3f800001, %1\n v_add_f32 %2, 0x3f800001, %2\n v_add_f32 %3, 0x3f800001, %3\n"

__global__ __launch_bounds__(512) void straight(float* out, int passes) {
  float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
  for (int p = 0; p < passes; ++p) {
    asm volatile(REP4096(I4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));   // 16384 instructions, 128 KB
  }
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(512) void straight_small(float* out, int passes) {   // 4096 instructions, 32 KB: a stage kernel's size
  float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
  for (int p = 0; p < passes; ++p) {
    asm volatile(REP256(I4) REP256(I4) REP256(I4) REP256(I4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(512) void looped(float* out, int trips) {
  float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
  for (int p = 0; p < trips; ++p) {
    asm volatile(REP16(I4) REP16(I4) REP16(I4) REP16(I4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));   // 256 instructions, 2 KB
  }
  out[blockIdx.x * 512 + threadIdx.x] = a + b + c + d;
}

template <class F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 512 * 4);
  // evict the instruction caches between launches with a big other kernel?  Launch boundaries already invalidate them.
  auto row = [&](const char* name, float ms, double instr) {
    std::printf("%-58s %8.1f us   %6.2f ns per wave-instruction stream position (%.0f instructions per wave)\n", name, ms * 1e3,
                ms * 1e6 / instr, instr);
  };
  row("straight 128 KB, executed once", time_ms([&] { hipLaunchKernelGGL(straight, dim3(512), dim3(512), 0, 0, out, 1); }, 5), 16384);
  row("straight 128 KB, two passes in one launch", time_ms([&] { hipLaunchKernelGGL(straight, dim3(512), dim3(512), 0, 0, out, 2); }, 5), 32768);
  row("straight 32 KB, executed once", time_ms([&] { hipLaunchKernelGGL(straight_small, dim3(512), dim3(512), 0, 0, out, 1); }, 5), 4096);
  row("straight 32 KB, two passes", time_ms([&] { hipLaunchKernelGGL(straight_small, dim3(512), dim3(512), 0, 0, out, 2); }, 5), 8192);
  row("straight 32 KB, four passes", time_ms([&] { hipLaunchKernelGGL(straight_small, dim3(512), dim3(512), 0, 0, out, 4); }, 5), 16384);
  row("looped 2 KB body x 16 trips (4096 instructions)", time_ms([&] { hipLaunchKernelGGL(looped, dim3(512), dim3(512), 0, 0, out, 16); }, 5), 4096);
  row("looped 2 KB body x 64 trips (16384 instructions)", time_ms([&] { hipLaunchKernelGGL(looped, dim3(512), dim3(512), 0, 0, out, 64); }, 5), 16384);
  row("empty-ish: looped x 0 trips", time_ms([&] { hipLaunchKernelGGL(looped, dim3(512), dim3(512), 0, 0, out, 0); }, 5), 1);
  return 0;
}
