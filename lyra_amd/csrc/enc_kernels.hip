// enc_kernels.hip -- encoder stages 0 and 1 as kernels of their own (bodies: enc_stages.h).
#include "enc_stages.h"

#ifdef LYRA_TIMING
extern "C" int lyra_hip_debug_exit_at_s0(int i) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lyra_exit_at), &i, sizeof(int));
}
extern "C" int lyra_hip_debug_wgtrace_s0(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_wgtrace), sizeof(long long) * 2048 * 4);
}
extern "C" int lyra_hip_debug_timing(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_tdbg), sizeof(long long) * 128);
}
#endif

#ifndef LYRA_C64_WAVES
#define LYRA_C64_WAVES 4   // waves per SIMD the 64-channel stage kernels are compiled for (3 -> up to 168 VGPRs, no spills)
#endif

namespace lyra {

namespace {
#ifndef LYRA_S0_STREAMS
#define LYRA_S0_STREAMS 4
#endif
constexpr int S0 = LYRA_S0_STREAMS;      // streams per workgroup (4 with 256 threads, 8 with 512)
}  // namespace

size_t enc_s0_lds_bytes() { return enc_s0_lds(S0); }
int enc_s0_streams_per_wg() { return S0; }
int enc_s0_threads() { return 64 * S0; }
size_t enc_s1_lds_bytes() { return enc_s1_lds(); }
int enc_s1_streams_per_wg() { return S1; }
int enc_s1_threads() { return NT1; }

__global__ __launch_bounds__(64 * S0, LYRA_C64_WAVES) void enc_s0_kernel(const EncS0P* __restrict__ Pp, const int16_t* __restrict__ pcm,
                                                           const int32_t* __restrict__ ids, int B,
                                                           uint8_t* __restrict__ state, float* __restrict__ out0,
                                                           int code_bytes, int tile0) {
  LYRA_STRESS(0);
  if (((int)blockIdx.x + tile0) * S0 >= B) return;
  enc_s0_body<S0>(*Pp, pcm, ids, B, state, out0, code_bytes, (int)blockIdx.x + tile0);
}

__global__ __launch_bounds__(NT1, NT1 == 512 ? 4 : 3) void enc_s1_kernel(const EncS1P* __restrict__ Pp, const float* __restrict__ in0,
                                                                        const int32_t* __restrict__ ids, int B,
                                                                        uint8_t* __restrict__ state, float* __restrict__ out1,
                                                                        int code_bytes, int tile0) {
  LYRA_STRESS(1);
  if (((int)blockIdx.x + tile0) * S1 >= B) return;
  enc_s1_body(*Pp, in0, ids, B, state, out1, code_bytes, (int)blockIdx.x + tile0);
}

}  // namespace lyra
