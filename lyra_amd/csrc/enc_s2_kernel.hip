// enc_s2_kernel.hip -- encoder stage 2 as a kernel of its own (body: enc_s2_stage.h).
#include "enc_s2_stage.h"

#ifdef LYRA_TIMING
extern "C" int lyra_hip_debug_exit_at_s2(int i) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lyra_exit_at), &i, sizeof(int));
}
extern "C" int lyra_hip_debug_timing_s2(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_tdbg), sizeof(long long) * 128);
}
extern "C" int lyra_hip_debug_wgtrace_s2(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_wgtrace), sizeof(long long) * 2048 * 4);
}
#endif

#ifndef LYRA_I8_WAVES
#define LYRA_I8_WAVES 4   // waves per SIMD the int8 stage kernels are compiled for (5 -> at most 96 VGPRs)
#endif
#ifndef LYRA_E2XN_WAVES
#define LYRA_E2XN_WAVES LYRA_I8_WAVES   // ... and the xnnpack-mode kernel on its own
#endif

namespace lyra {

size_t enc_s2_lds_bytes() { return enc_s2_lds(); }
int enc_s2_streams_per_wg() { return S2; }

__global__ __launch_bounds__(NT2, LYRA_I8_WAVES) void enc_s2_kernel(const EncS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ feats,
                                                      float* __restrict__ codes_dbg, int code_bytes, int tile0) {
  LYRA_STRESS(2);
  if (((int)blockIdx.x + tile0) * S2 >= B) return;
  enc_s2_body<0>(Pp, in1, ids, B, state, feats, codes_dbg, code_bytes, (int)blockIdx.x + tile0);
}
__global__ __launch_bounds__(NT2, LYRA_I8_WAVES) void enc_s2_dr_kernel(const EncS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                         const int32_t* __restrict__ ids, int B,
                                                         uint8_t* __restrict__ state, float* __restrict__ feats,
                                                         float* __restrict__ codes_dbg, int code_bytes, int tile0) {
  LYRA_STRESS(2);
  if (((int)blockIdx.x + tile0) * S2 >= B) return;
  enc_s2_body<1>(Pp, in1, ids, B, state, feats, codes_dbg, code_bytes, (int)blockIdx.x + tile0);
}
// mode 3 "builtin_mixed": TFLite's builtin int8 kernels per operator (lyra_dev.h conv_flavour)
__global__ __launch_bounds__(NT2, LYRA_I8_WAVES) void enc_s2_bm_kernel(const EncS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                         const int32_t* __restrict__ ids, int B,
                                                         uint8_t* __restrict__ state, float* __restrict__ feats,
                                                         float* __restrict__ codes_dbg, int code_bytes, int tile0) {
  LYRA_STRESS(2);
  if (((int)blockIdx.x + tile0) * S2 >= B) return;
  enc_s2_body<3>(Pp, in1, ids, B, state, feats, codes_dbg, code_bytes, (int)blockIdx.x + tile0);
}
// mode 2 "xnnpack" (the default): XNNPACK's QS8 arithmetic
__global__ __launch_bounds__(NT2, LYRA_E2XN_WAVES) void enc_s2_xn_kernel(const EncS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                         const int32_t* __restrict__ ids, int B,
                                                         uint8_t* __restrict__ state, float* __restrict__ feats,
                                                         float* __restrict__ codes_dbg, int code_bytes, int tile0) {
  LYRA_STRESS(2);
  if (((int)blockIdx.x + tile0) * S2 >= B) return;
#ifdef LYRA_I8_PRIO   // experiment: the int8 stages are latency chains with little issue demand -- let them go first
  __builtin_amdgcn_s_setprio(LYRA_I8_PRIO);
#endif
  enc_s2_body<2>(Pp, in1, ids, B, state, feats, codes_dbg, code_bytes, (int)blockIdx.x + tile0);
}

}  // namespace lyra
