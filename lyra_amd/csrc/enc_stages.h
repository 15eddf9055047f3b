// enc_stages.h -- SoundStream encoder stages 0 and 1 as device functions (soundstream_encoder.tflite ops 15-93 as run
// by SoundStreamEncoder::Extract, lyra/soundstream_encoder.cc:53-64).  Stage 2 is in enc_s2_stage.h.
//
//   enc_s0  S0 streams/WG  PCM -> first conv k64/s16 -> 3 resblocks @64ch x 20 rows -> conv k10/s5   (fp32)
//   enc_s1  8 streams/WG   3 resblocks @128ch x 4 rows (2nd conv g=2) -> conv k4/s2 g=2               (fp32)
//
// Every fp32 dot product runs on v_mfma_f32_16x16x4_f32 in ascending-k order starting from the bias (== the oracle's
// fmaf chain == XNNPACK's micro-kernel order);
// everything between two GEMMs (LeakyReLU, depthwise dilated conv, residual add, history update) is fused
// around them in LDS/registers.  The stages are launched as kernels of their own (enc_kernels.hip: stage 0 with
// 4 streams / 256 threads / 29 KB LDS, four tiles per CU; stage 1 with 8 streams / 512 threads / 35 KB, two per CU) or
// back to back on one 8-stream / 512-thread tile by the per-side kernel (enc_side_kernel.hip).
// Per stream and step the only HBM traffic is PCM in, history read/write and one small inter-stage activation.
#ifndef LYRA_AMD_CSRC_ENC_STAGES_H_
#define LYRA_AMD_CSRC_ENC_STAGES_H_
#include "resblocks.h"

namespace lyra {

namespace {
constexpr int CS0 = 72;    // LDS row stride (64 + 8) floats
constexpr int PBS = 376;   // PCM staging row stride (368 + 8) floats
constexpr int S1 = 8;
constexpr int CS1 = 136;   // 128 + 8
#ifndef LYRA_S1_THREADS
#define LYRA_S1_THREADS 512   // 8 waves per tile: 4 waves per SIMD with two tiles per CU (256 = the 4-wave layout)
#endif
constexpr int NT1 = LYRA_S1_THREADS;
constexpr int NW1 = NT1 / 64;
}  // namespace

__host__ __device__ constexpr size_t enc_s0_lds(int s0) { return (size_t)(25 * s0 * CS0) * 4 + 64; }
__host__ __device__ constexpr size_t enc_s1_lds() { return (size_t)(6 * S1 * CS1 + 4 * S1 * CS1 + 4 * S1 * CS1) * 4 + 2 * S1 * 4; }

// =============================================================================================
// stage 0: S0 streams per workgroup of 64 * S0 threads (4 / 256 or 8 / 512)
// =============================================================================================
template <int S0>
__device__ __forceinline__ void enc_s0_body(const EncS0P& P, const int16_t* __restrict__ pcm,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            float* __restrict__ out0, int code_bytes, int tile = (int)blockIdx.x) {
  constexpr int NT0 = 64 * S0, NW0 = NT0 / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                     // [25][S0][CS0]: rows 0-4 strided-conv history, rows 5-24 the one
                                        // activation matrix of the residual blocks; first: PCM staging
  int* sids = reinterpret_cast<int*>(XB + 25 * S0 * CS0);
  static_assert(S0 * PBS <= 25 * S0 * CS0, "PCM staging fits");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = tile * S0;
  LYRA_WG_BEGIN();
  LYRA_TSTAMP(0);
  LYRA_WSTAMP(100);
  wg_schedule_hint();
  if (tid < S0) sids[tid] = ids[min(b0 + tid, B - 1)];
  const auto warm = l2_warm<NT0, 1>(P.warm);
  const auto warm_code = code_warm<NT0>(code_bytes);
  LYRA_SYNC_KEEP();
  // uniform base + 32-bit per-lane offset (TileCtx::at): scalar-base global accesses, 32-bit address arithmetic
  auto soff = [&](int s) -> uint32_t { return (uint32_t)max(sids[s], 0) * (uint32_t)st::E0_BYTES; };
  auto gat = [&](uint32_t o) -> uint8_t LYRA_GLOBAL* { return (uint8_t LYRA_GLOBAL*)state + o; };
  auto valid = [&](int s) -> bool { return b0 + s < B && sids[s] >= 0; };   // id -1 = masked slot (TileCtx::valid)

  // The 5 history rows of the strided conv (needed only in phase D/E) are requested together with the PCM so
  // that their HBM latency is paid once, up front; they are parked in registers until the staging area is free.
  f32x4 halo[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * NT0;
    const int p4 = idx & 15, s = (idx >> 4) & (S0 - 1), j = (idx >> 4) / S0;
    halo[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (idx < 5 * S0 * 16) halo[k] = *(const f32x4 LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_D0 + (j * 64 + p4 * 4) * 4));
  }

  // ---- A. window = [48 history samples | 320 new samples] / 32768, AT16 order ----------------
  float* PB = XB;
  for (int idx = tid; idx < S0 * 46; idx += NT0) {   // 46 = 6 history + 40 PCM pieces of 8 samples
    int s = idx / 46, v = idx - s * 46;
    float x[8];
    if (v < 6) {
      const f32x4 LYRA_GLOBAL* h = (const f32x4 LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_FIRST + v * 32));
      f32x4 a = h[0], b = h[1];
      x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
    } else {
      int sb = min(s, B - 1 - b0);
      i32x4 w = *goff<const i32x4>(pcm + (size_t)b0 * 320, (uint32_t)(sb * 640 + (v - 6) * 16));
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // Int16ToUnitScalar, dsp_utils.h:106-108
        x[2 * e] = (float)(int16_t)(w[e] & 0xffff) * (1.0f / 32768.0f);
        x[2 * e + 1] = (float)(int16_t)(w[e] >> 16) * (1.0f / 32768.0f);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) PB[s * PBS + at16(v * 8 + e)] = x[e];
  }
  __syncthreads();
  for (int idx = tid; idx < S0 * 48; idx += NT0) {
    int s = idx / 48, i = idx - s * 48;
    if (valid(s)) *(float LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_FIRST + i * 4)) = PB[s * PBS + at16(320 + i)];
  }
  LYRA_TSTAMP(1);

  const int wn = wave & 3, wm = wave >> 2;  // GEMM wave grid for N = 64: N tile x M group of 5 tiles
  const int ncol = wn * 16 + (lane & 15);   // logical output channel of this lane's C column
  const int pcol = at16(ncol);
  constexpr bool SW = LYRA_SWAP64 != 0;     // operand-swapped GEMMs: a lane holds 4 consecutive channels of ONE row (lyra_dev.h)

  // ---- B. first conv k64/s16: [20*S rows] x K=64 x N=64 ---------------------------------------
  f32x4 xr[5][1];  // the residual stream X, resident in registers (MFMA C layout) through the three blocks
  {
    auto aoff = [&](int i, int c) {
      int R = (wm * 5 + i) * 16 + m;
      return (R & (S0 - 1)) * PBS + (R / S0 + c) * 16 + q * 4;
    };
    gemm_f32_bias<5, 1, 4, 4, gemm_pf<5, 1>(), SW>(PB, aoff, P.first.w + wn * 4 * 64, P.first.b, wn * 16, xr);
  }
  __syncthreads();  // PCM staging area is free again
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * NT0;
    const int p4 = idx & 15, s = (idx >> 4) & (S0 - 1), j = (idx >> 4) / S0;
    if (idx < 5 * S0 * 16) *reinterpret_cast<f32x4*>(&XB[(j * S0 + s) * CS0 + p4 * 4]) = halo[k];
  }
  LYRA_TSTAMP(2);

  // ---- C. three residual blocks, dilation 1 / 3 / 9 --------------------------------------------
  TileCtx cx{state, sids, nullptr, B - b0, st::E0_BYTES};
  resblocks64r<S0, NT0>(xr, XB + 5 * S0 * CS0, cx, P.dw, P.pw, P.cv, st::E_R0_0, st::E_R0_1, st::E_R0_2);
  LYRA_TSTAMP(3);

  // ---- D. a = lrelu(X) -> rows 5..24 (rows 0-4 already hold the strided conv's history) ----------------
#pragma unroll
  for (int i = 0; i < 5; ++i)
{
    const f32x4 a4 = lrelu4(xr[i][0]);
    if constexpr (SW) {
      *reinterpret_cast<f32x4*>(&XB[(5 * S0 + (wm * 5 + i) * 16 + m) * CS0 + wn * 16 + q * 4]) = a4;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) XB[(5 * S0 + (wm * 5 + i) * 16 + q * 4 + e) * CS0 + pcol] = a4[e];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 5 * S0 * 16; idx += NT0) {
    int p4 = idx & 15, s = (idx >> 4) & (S0 - 1), j = (idx >> 4) / S0;
    if (valid(s))
      *(f32x4 LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_D0 + (j * 64 + p4 * 4) * 4)) =
          *reinterpret_cast<const f32x4*>(&XB[((20 + j) * S0 + s) * CS0 + p4 * 4]);
  }
  LYRA_TSTAMP(4);

  // ---- E. conv k10/s5: [4*S rows] x K=640 x N=128 -----------------------------------------------
  {
    constexpr int MTW = (4 * S0) / 16, NTW = 8 / NW0;
    f32x4 acc[MTW][NTW];
    auto aoff = [&](int i, int c) {
      int tap = c >> 2, c16 = c & 3;
      int R = i * 16 + m;
      return ((5 * (R / S0) + tap) * S0 + (R & (S0 - 1))) * CS0 + c16 * 16 + q * 4;
    };
    gemm_f32_bias<MTW, NTW, 40, 40, gemm_pf<MTW, NTW>(), SW>(XB, aoff, P.down.w + (wave * NTW) * 40 * 64, P.down.b,
                                                             wave * NTW * 16, acc);
    LYRA_TSTAMP(5);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if constexpr (SW) {   // row (tau, s) = this lane's C column, 4 consecutive physical channels: one 16-byte store
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int R = i * 16 + m, tau = R / S0, s = R & (S0 - 1);
          if (valid(s))
            *goff<f32x4>(out0 + (size_t)b0 * 512, (uint32_t)(((s * 4 + tau) * 128 + (wave * NTW + j) * 16 + q * 4) * 4)) = acc[i][j];
        }
      } else {
        int n = (wave * NTW + j) * 16 + (lane & 15);
        int pc = at16(n);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int R = i * 16 + q * 4 + e, tau = R / S0, s = R & (S0 - 1);
            if (valid(s)) *goff<float>(out0 + (size_t)b0 * 512, (uint32_t)(((s * 4 + tau) * 128 + pc) * 4)) = acc[i][j][e];
          }
      }
    }
  }
  LYRA_WG_END();
  LYRA_TSTAMP(6);
  LYRA_WSTAMP(101);
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
}

// =============================================================================================
// stage 1
// =============================================================================================
__device__ __forceinline__ void enc_s1_body(const EncS1P& P, const float* __restrict__ in0,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            float* __restrict__ out1, int code_bytes, int tile = (int)blockIdx.x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                    // [6][S1][CS1]: rows 0-1 strided-conv history, rows 2-5 X[t]
  float* DB = XB + 6 * S1 * CS1;       // [4][S1][CS1]
  float* PB = DB + 4 * S1 * CS1;       // [4][S1][CS1]
  int* sids = reinterpret_cast<int*>(PB + 4 * S1 * CS1);
  int* sphase = sids + S1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = tile * S1;
  wg_schedule_hint();
  LYRA_TSTAMP(70);
  LYRA_WSTAMP(102);
  // The stage input does not depend on the stream ids (only on the tile): requested in the same round trip as the ids, in
  // front of the barrier the ids sit behind -- one dependent memory trip less at the head of the kernel's chain.
  int my_id = 0;
  if (tid < S1) my_id = ids[min(b0 + tid, B - 1)];
  constexpr int XIN = (4 * S1 * 32) / NT1;
  static_assert((4 * S1 * 32) % NT1 == 0 && (4 * S1 * 32) / NT1 >= 1, "the unrolled input prefetch covers the stage input only when the thread count divides it");
  f32x4 xin[XIN];
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NT1;
    int p4 = idx & 31, s = (idx >> 5) & (S1 - 1), t = (idx >> 5) / S1;
    int sb = min(s, B - 1 - b0);
    xin[k] = *goff<const f32x4>(in0 + (size_t)b0 * 512, (uint32_t)(((sb * 4 + t) * 128 + p4 * 4) * 4));
  }
  if (tid < S1) {
    sids[tid] = my_id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)max(my_id, 0) * st::E1_BYTES + st::PHASE);
  }
  const auto warm = l2_warm<NT1, 2>(P.warm);
  const auto warm_code = code_warm<NT1>(code_bytes);
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NT1;
    int p4 = idx & 31, s = (idx >> 5) & (S1 - 1), t = (idx >> 5) / S1;
    *reinterpret_cast<f32x4*>(&XB[((2 + t) * S1 + s) * CS1 + p4 * 4]) = xin[k];
  }
  LYRA_SYNC_KEEP();
  auto soff = [&](int s) -> uint32_t { return (uint32_t)max(sids[s], 0) * (uint32_t)st::E1_BYTES; };
  auto gat = [&](uint32_t o) -> uint8_t LYRA_GLOBAL* { return (uint8_t LYRA_GLOBAL*)state + o; };
  auto valid = [&](int s) -> bool { return b0 + s < B && sids[s] >= 0; };

  TileCtx cx{state, sids, sphase, B - b0, st::E1_BYTES};
  const auto H0 = hist128_prefetch<S1, NT1>(cx, 1, st::E_R1_0);   // first block's history: same round trip as the input
  for (int idx = tid; idx < 2 * S1 * 32; idx += NT1) {   // strided conv's 2 history rows
    int p4 = idx & 31, s = (idx >> 5) & (S1 - 1), j = (idx >> 5) / S1;
    *reinterpret_cast<f32x4*>(&XB[(j * S1 + s) * CS1 + p4 * 4]) =
        *(const f32x4 LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_D1 + (j * 128 + p4 * 4) * 4));
  }
  __syncthreads();

  LYRA_TSTAMP(71);
  resblocks128<S1, NT1>(XB + 2 * S1 * CS1, DB, PB, cx, P.dw, P.pw, P.cv, st::E_R1_0, st::E_R1_1, st::E_R1_2, H0);

  LYRA_TSTAMP(72);
  for (int idx = tid; idx < 4 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, rs = idx >> 5;
    f32x4* x = reinterpret_cast<f32x4*>(&XB[(2 * S1 + rs) * CS1 + p4 * 4]);
    *x = lrelu4(*x);
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, s = (idx >> 5) & (S1 - 1), j = (idx >> 5) / S1;
    if (valid(s))
      *(f32x4 LYRA_GLOBAL*)gat(soff(s) + (uint32_t)(st::E_D1 + (j * 128 + p4 * 4) * 4)) =
          *reinterpret_cast<const f32x4*>(&XB[((4 + j) * S1 + s) * CS1 + p4 * 4]);
  }

  {  // conv k4/s2, 2 groups: per group [2*S rows] x K=256 x N=128; a wave's N tiles lie in one group
    constexpr int MTW = (2 * S1) / 16, NTW = 16 / NW1;
    f32x4 acc[MTW][NTW];
    const int nt0 = wave * NTW, g = nt0 >> 3;
    auto aoff = [&](int i, int c) {
      int tap = c >> 2, c16 = c & 3;
      int R = i * 16 + m;
      return ((2 * (R / S1) + tap) * S1 + (R & (S1 - 1))) * CS1 + g * 64 + c16 * 16 + q * 4;
    };
    constexpr bool SW = LYRA_SWAP128 != 0;   // operand-swapped (lyra_dev.h): one 16-byte store per C tile
    gemm_f32_bias<MTW, NTW, 16, 16, gemm_pf<MTW, NTW>(), SW>(XB, aoff, P.down.w + nt0 * 16 * 64, P.down.b, nt0 * 16, acc);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if constexpr (SW) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int R = i * 16 + m, tau = R / S1, s = R & (S1 - 1);
          if (valid(s))
            *goff<f32x4>(out1 + (size_t)b0 * 512, (uint32_t)(((s * 2 + tau) * 256 + (nt0 + j) * 16 + q * 4) * 4)) = acc[i][j];
        }
      } else {
        int n = (nt0 + j) * 16 + (lane & 15);
        int pc = at16(n);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int R = i * 16 + q * 4 + e, tau = R / S1, s = R & (S1 - 1);
            if (valid(s)) *goff<float>(out1 + (size_t)b0 * 512, (uint32_t)(((s * 2 + tau) * 256 + pc) * 4)) = acc[i][j][e];
          }
      }
    }
  }
  LYRA_TSTAMP(73);
  LYRA_WSTAMP(103);
  if (tid < S1 && valid(tid)) {   // this region's ring phase (every thread read it into LDS before the first barrier)
    int ph = sphase[tid] + 1;
    *(int LYRA_GLOBAL*)gat(soff(tid) + (uint32_t)st::PHASE) = ph >= st::PHASE_MOD ? 0 : ph;
  }
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
}

}  // namespace lyra
#endif
