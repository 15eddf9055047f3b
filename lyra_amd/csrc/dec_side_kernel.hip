// dec_side_kernel.hip -- the whole decoder side (LyraGanModel::RunConditioning + RunModel, lyra/lyra_gan_model.cc:53-64:
// ONE Invoke() of lyragan.tflite) as ONE launch: a workgroup of 512 threads takes a tile of 8 streams through stage 0,
// stage 1 and stage 2 back to back (bodies: dec_stages.h).  See enc_side_kernel.hip for what this buys; the
// inter-stage activations travel through d_d0 / d_d1, written and read by the same workgroup.
#include "dec_stages.h"

namespace lyra {

namespace {
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
}  // namespace

size_t dec_side_lds_bytes() { return cmax(cmax(dec_s0_lds(), dec_s1_lds()), dec_s2_lds(8)); }

template <int MODE>
__device__ __forceinline__ void dec_side_body(const DecS0P* __restrict__ P0, const DecS1P* __restrict__ P1,
                                              const DecS2P* __restrict__ P2, const float* __restrict__ feats,
                                              const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ st0,
                                              uint8_t* __restrict__ st1, uint8_t* __restrict__ st2,
                                              float* __restrict__ d0, float* __restrict__ d1,
                                              int16_t* __restrict__ pcm, const uint8_t* __restrict__ packets,
                                              int num_stages, const float* __restrict__ cb, int code_bytes) {
  static_assert(SD0 == 8 && SD1 == 8 && NTD0 == 512 && NTD1 == 512, "one 8-stream / 512-thread tile through all stages");
  dec_s0_body<MODE>(P0, feats, ids, B, st0, d0, packets, num_stages, cb, code_bytes);
  __syncthreads();   // d0 written (vmcnt drained by the barrier's fence); LDS free for the next stage
  dec_s1_body(*P1, d0, ids, B, st1, d1, 0);
  __syncthreads();
  dec_s2_body<8>(*P2, d1, ids, B, st2, pcm, 0);
}

__global__ __launch_bounds__(512, 4) void dec_side_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2,
                                                         const float* feats, const int32_t* ids, int B, uint8_t* st0,
                                                         uint8_t* st1, uint8_t* st2, float* d0, float* d1, int16_t* pcm,
                                                         const uint8_t* packets, int num_stages, const float* cb,
                                                         int code_bytes) {
  dec_side_body<0>(P0, P1, P2, feats, ids, B, st0, st1, st2, d0, d1, pcm, packets, num_stages, cb, code_bytes);
}
__global__ __launch_bounds__(512, 4) void dec_side_dr_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2,
                                                            const float* feats, const int32_t* ids, int B, uint8_t* st0,
                                                            uint8_t* st1, uint8_t* st2, float* d0, float* d1,
                                                            int16_t* pcm, const uint8_t* packets, int num_stages,
                                                            const float* cb, int code_bytes) {
  dec_side_body<1>(P0, P1, P2, feats, ids, B, st0, st1, st2, d0, d1, pcm, packets, num_stages, cb, code_bytes);
}
__global__ __launch_bounds__(512, 4) void dec_side_xn_kernel(const DecS0P* P0, const DecS1P* P1, const DecS2P* P2,
                                                            const float* feats, const int32_t* ids, int B, uint8_t* st0,
                                                            uint8_t* st1, uint8_t* st2, float* d0, float* d1,
                                                            int16_t* pcm, const uint8_t* packets, int num_stages,
                                                            const float* cb, int code_bytes) {
  dec_side_body<2>(P0, P1, P2, feats, ids, B, st0, st1, st2, d0, d1, pcm, packets, num_stages, cb, code_bytes);
}

// Stages 0 + 1 only (both are 8-stream / 512-thread tiles already)
size_t dec_s01_lds_bytes() { return cmax(dec_s0_lds(), dec_s1_lds()); }
__global__ __launch_bounds__(512, 4) void dec_s01_xn_kernel(const DecS0P* P0, const DecS1P* P1, const float* feats, const int32_t* ids,
                                                           int B, uint8_t* st0, uint8_t* st1, float* d0, float* d1,
                                                           const uint8_t* packets, int num_stages, const float* cb, int code_bytes) {
  dec_s0_body<2>(P0, feats, ids, B, st0, d0, packets, num_stages, cb, code_bytes);
  __syncthreads();
  dec_s1_body(*P1, d0, ids, B, st1, d1, 0);
}

}  // namespace lyra
