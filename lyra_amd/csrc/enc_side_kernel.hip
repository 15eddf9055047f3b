// enc_side_kernel.hip -- the whole encoder side (SoundStreamEncoder::Extract, lyra/soundstream_encoder.cc:53-64: ONE
// Invoke() of soundstream_encoder.tflite) as ONE launch: a workgroup of 512 threads takes a tile of 8 streams through
// stage 0, stage 1 and stage 2 back to back (bodies: enc_stages.h, enc_s2_stage.h).  Compared with three launches this
// removes two kernel boundaries (~5 us each: drain, L2 write-back / invalidate, dispatch), two per-kernel tails (the
// slowest tile of a one-wave grid ends ~6 us after the median one; in the chained form a tile that is late in one
// stage simply starts the next one late) and keeps the inter-stage activations of a tile in that CU's caches.
// The activations still travel through the d_e0 / d_e1 buffers (written and read by the same workgroup: a workgroup
// barrier orders them, all its waves share one L1), so the stage bodies are the ones the per-stage kernels run.
#include "enc_s2_stage.h"
#include "enc_stages.h"

namespace lyra {

namespace {
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
}  // namespace

size_t enc_side_lds_bytes() { return cmax(cmax(enc_s0_lds(8), enc_s1_lds()), enc_s2_lds()); }

template <int MODE>
__device__ __forceinline__ void enc_side_body(const EncS0P* __restrict__ P0, const EncS1P* __restrict__ P1,
                                              const EncS2P* __restrict__ P2, const int16_t* __restrict__ pcm,
                                              const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ st0,
                                              uint8_t* __restrict__ st1, uint8_t* __restrict__ st2,
                                              float* __restrict__ e0, float* __restrict__ e1,
                                              float* __restrict__ feats, float* __restrict__ codes_dbg, int code_bytes) {
  static_assert(S1 == 8 && S2 == 8 && NT1 == 512 && NT2 == 512, "one 8-stream / 512-thread tile through all stages");
  enc_s0_body<8>(*P0, pcm, ids, B, st0, e0, code_bytes);
  __syncthreads();   // e0 written (vmcnt drained by the barrier's fence); LDS free for the next stage
  enc_s1_body(*P1, e0, ids, B, st1, e1, 0);
  __syncthreads();
  enc_s2_body<MODE>(P2, e1, ids, B, st2, feats, codes_dbg, 0);
}

__global__ __launch_bounds__(512, 4) void enc_side_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2,
                                                         const int16_t* pcm, const int32_t* ids, int B, uint8_t* st0,
                                                         uint8_t* st1, uint8_t* st2, float* e0, float* e1,
                                                         float* feats, float* codes_dbg, int code_bytes) {
  enc_side_body<0>(P0, P1, P2, pcm, ids, B, st0, st1, st2, e0, e1, feats, codes_dbg, code_bytes);
}
__global__ __launch_bounds__(512, 4) void enc_side_dr_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2,
                                                            const int16_t* pcm, const int32_t* ids, int B, uint8_t* st0,
                                                            uint8_t* st1, uint8_t* st2, float* e0, float* e1,
                                                            float* feats, float* codes_dbg, int code_bytes) {
  enc_side_body<1>(P0, P1, P2, pcm, ids, B, st0, st1, st2, e0, e1, feats, codes_dbg, code_bytes);
}
__global__ __launch_bounds__(512, 4) void enc_side_xn_kernel(const EncS0P* P0, const EncS1P* P1, const EncS2P* P2,
                                                            const int16_t* pcm, const int32_t* ids, int B, uint8_t* st0,
                                                            uint8_t* st1, uint8_t* st2, float* e0, float* e1,
                                                            float* feats, float* codes_dbg, int code_bytes) {
  enc_side_body<2>(P0, P1, P2, pcm, ids, B, st0, st1, st2, e0, e1, feats, codes_dbg, code_bytes);
}

// Stages 1 + 2 only (both are 8-stream / 512-thread tiles already): two kernel boundaries per side instead of three.
size_t enc_s12_lds_bytes() { return cmax(enc_s1_lds(), enc_s2_lds()); }
__global__ __launch_bounds__(512, 4) void enc_s12_xn_kernel(const EncS1P* P1, const EncS2P* P2, const float* e0, const int32_t* ids,
                                                           int B, uint8_t* st1, uint8_t* st2, float* e1, float* feats,
                                                           float* codes_dbg, int code_bytes) {
  enc_s1_body(*P1, e0, ids, B, st1, e1, code_bytes);
  __syncthreads();
  enc_s2_body<2>(P2, e1, ids, B, st2, feats, codes_dbg, 0);
}

}  // namespace lyra
