// dec_kernels.hip -- the three decoder stages as kernels of their own (bodies: dec_stages.h).
#include "dec_stages.h"

#ifdef LYRA_TIMING
extern "C" int lyra_hip_debug_exit_at_d0(int i) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lyra_exit_at), &i, sizeof(int));
}
extern "C" int lyra_hip_debug_timing_d0(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_tdbg), sizeof(long long) * 128);
}
extern "C" int lyra_hip_debug_wgtrace_d0(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_wgtrace), sizeof(long long) * 2048 * 4);
}
#endif

#ifndef LYRA_I8_WAVES
#define LYRA_I8_WAVES 4   // waves per SIMD the int8 stage kernels are compiled for (5 -> at most 96 VGPRs)
#endif
#ifndef LYRA_D0XN_WAVES
#define LYRA_D0XN_WAVES LYRA_I8_WAVES   // ... and the xnnpack-mode kernel on its own
#endif

#ifndef LYRA_C64_WAVES
#define LYRA_C64_WAVES 4   // waves per SIMD the 64-channel stage kernels are compiled for (3 -> up to 168 VGPRs, no spills)
#endif

namespace lyra {

size_t dec_s0_lds_bytes() { return dec_s0_lds(); }
int dec_s0_streams_per_wg() { return SD0; }

__global__ __launch_bounds__(NTD0, LYRA_I8_WAVES) void dec_s0_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                       const int32_t* __restrict__ ids, int B,
                                                       uint8_t* __restrict__ state, float* __restrict__ out0,
                                                       const uint8_t* __restrict__ packets, int num_stages,
                                                       const float* __restrict__ cb, int code_bytes, int tile0) {
  LYRA_STRESS(4);
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
  dec_s0_body<0>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}
__global__ __launch_bounds__(NTD0, LYRA_I8_WAVES) void dec_s0_dr_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                          const int32_t* __restrict__ ids, int B,
                                                          uint8_t* __restrict__ state, float* __restrict__ out0,
                                                          const uint8_t* __restrict__ packets, int num_stages,
                                                          const float* __restrict__ cb, int code_bytes, int tile0) {
  LYRA_STRESS(4);
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
  dec_s0_body<1>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}
// mode 3 "builtin_mixed": TFLite's builtin int8 kernels per operator (lyra_dev.h conv_flavour)
__global__ __launch_bounds__(NTD0, LYRA_I8_WAVES) void dec_s0_bm_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                          const int32_t* __restrict__ ids, int B,
                                                          uint8_t* __restrict__ state, float* __restrict__ out0,
                                                          const uint8_t* __restrict__ packets, int num_stages,
                                                          const float* __restrict__ cb, int code_bytes, int tile0) {
  LYRA_STRESS(4);
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
  dec_s0_body<3>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}
// mode 2 "xnnpack" (the default): XNNPACK's QS8 arithmetic
__global__ __launch_bounds__(NTD0, LYRA_D0XN_WAVES) void dec_s0_xn_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                          const int32_t* __restrict__ ids, int B,
                                                          uint8_t* __restrict__ state, float* __restrict__ out0,
                                                          const uint8_t* __restrict__ packets, int num_stages,
                                                          const float* __restrict__ cb, int code_bytes, int tile0) {
  LYRA_STRESS(4);
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
#ifdef LYRA_I8_PRIO
  __builtin_amdgcn_s_setprio(LYRA_I8_PRIO);
#endif
  dec_s0_body<2>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}

size_t dec_s1_lds_bytes() { return dec_s1_lds(); }
int dec_s1_streams_per_wg() { return SD1; }
int dec_s1_threads() { return NTD1; }

__global__ __launch_bounds__(NTD1, NTD1 == 512 ? 4 : 3) void dec_s1_kernel(const DecS1P* __restrict__ Pp, const float* __restrict__ in0,
                                                                          const int32_t* __restrict__ ids, int B,
                                                                          uint8_t* __restrict__ state, float* __restrict__ out1,
                                                                          int code_bytes, int tile0) {
  LYRA_STRESS(5);
  if (((int)blockIdx.x + tile0) * SD1 >= B) return;
  dec_s1_body(*Pp, in0, ids, B, state, out1, code_bytes, (int)blockIdx.x + tile0);
}

namespace {
#ifndef LYRA_S0_STREAMS
#define LYRA_S0_STREAMS 4
#endif
constexpr int SD2 = LYRA_S0_STREAMS;   // 4 streams with 256 threads, or 8 with 512
}  // namespace

size_t dec_s2_lds_bytes() { return dec_s2_lds(SD2); }
int dec_s2_streams_per_wg() { return SD2; }
int dec_s2_threads() { return 64 * SD2; }

__global__ __launch_bounds__(64 * SD2, LYRA_C64_WAVES) void dec_s2_kernel(const DecS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                            const int32_t* __restrict__ ids, int B,
                                                            uint8_t* __restrict__ state, int16_t* __restrict__ pcm,
                                                            int code_bytes, int tile0) {
  LYRA_STRESS(6);
  if (((int)blockIdx.x + tile0) * SD2 >= B) return;
  dec_s2_body<SD2>(*Pp, in1, ids, B, state, pcm, code_bytes, (int)blockIdx.x + tile0);
}


}  // namespace lyra
