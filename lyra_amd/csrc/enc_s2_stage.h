// enc_s2_stage.h -- encoder stage 2 (256 channels x 2 rows per frame): resblock 0 with fp32 depthwise +
// pointwise and an int8 tail, two int8 resblocks, int8 conv k4/s2 g4 -> [1][512], int8 bottleneck conv k3 g4
// -> 64 int8 codes -> features (graph ops 94-151 of soundstream_encoder.tflite).
//
// The stage is a long chain of small dependent phases, latency- and VALU-bound rather than MFMA- or HBM-bound:
// the tile is kept small (S = 8 streams, 512 threads, ~54 KB LDS) so that two workgroups share a CU; int8
// LeakyReLU / ADD rescalings are LDS table lookups, the int8 residual blocks keep LeakyReLU -> depthwise ->
// history thread-local, and everything a block loads from global memory is requested one phase early
// (resblock_q.h).  Rows are (t, s) -> t*S + s; with S = 8 the two time steps fill exactly one 16-row MFMA tile.
#ifndef LYRA_AMD_CSRC_ENC_S2_STAGE_H_
#define LYRA_AMD_CSRC_ENC_S2_STAGE_H_
#include "resblock_q.h"

namespace lyra {

namespace {
constexpr int S2 = 8;
constexpr int CS2 = 264;   // 256 + 8 floats
constexpr int QS = 288;    // int8 row stride, C = 256
constexpr int QS5 = 544;   // int8 row stride, C = 512
constexpr int NT2 = 512;
constexpr int MT2 = (2 * S2) / 16;          // M tiles of a [2][S] matrix
constexpr int XF_BYTES = 2 * S2 * CS2 * 4;
constexpr int QB_BYTES = 2 * S2 * QS;
constexpr int NLR = 7, NADD = 2;            // LeakyReLU / ADD lookup tables (resblock_q.h)
static_assert(S2 == 8 || S2 == 16, "tile sizes the index math below supports");
static_assert(4 * S2 * QS + 16 * QS <= XF_BYTES && 3 * S2 * QS5 + 16 * QS5 <= XF_BYTES + 3 * QB_BYTES,
              "aliased int8 staging buffers (incl. the rows a partial M tile over-reads) must fit");
}  // namespace

__host__ __device__ constexpr size_t enc_s2_lds() { return (size_t)2 * XF_BYTES + 3 * QB_BYTES + 2 * S2 * 4 + NLR * 256 + NADD * 2048; }

// MODE: arithmetic flavour of the int8 region (0 exact / 1 gemmlowp double rounding / 2 xnnpack / 3 builtin_mixed), a compile-time constant:
// as a run-time value every requantisation carried both arithmetic paths and a (uniform) branch -- a third of this
// kernel's instructions -- which costs issue slots and, the code being executed once per workgroup, instruction fetch.
template <int MODE>
__device__ __forceinline__ void enc_s2_body(const EncS2P* __restrict__ Pp, const float* __restrict__ in1,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            float* __restrict__ feats, float* __restrict__ codes_dbg, int code_bytes, int tile = (int)blockIdx.x) {
  const EncS2P& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* DF = smem;                                  // [2][S][CS2] depthwise out; later int8 staging QB4
  float* XF = DF + 2 * S2 * CS2;                     // [2][S][CS2] stage input (float skip); later QC (with QX..)
  int8_t* QX = reinterpret_cast<int8_t*>(XF + 2 * S2 * CS2);  // residual stream (int8)
  int8_t* QD = QX + QB_BYTES;
  int8_t* QP = QD + QB_BYTES;
  int* sids = reinterpret_cast<int*>(QP + QB_BYTES);
  int* sphase = sids + S2;
  int32_t* LA = sphase + S2;                         // [NADD][2][256] ADD operand tables
  int8_t* LQ = reinterpret_cast<int8_t*>(LA + NADD * 512);  // [NLR][256] LeakyReLU tables
  int8_t* QB4 = reinterpret_cast<int8_t*>(DF);       // [4][S][QS]   (aliases DF once it is dead)
  int8_t* QC = reinterpret_cast<int8_t*>(XF);        // [3][S][QS5]  (aliases XF and, past it, QX.. once dead)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = tile * S2;
  wg_schedule_hint();
  LYRA_TSTAMP(0);
  LYRA_WSTAMP(100);
  LYRA_WG_BEGIN();
  // the stage input does not depend on the stream ids: requested with them, ahead of the barrier (see enc_s1_body)
  int my_id = 0;
  if (tid < S2) my_id = ids[min(b0 + tid, B - 1)];
  constexpr int XIN = (2 * S2 * 64) / NT2;
  static_assert((2 * S2 * 64) % NT2 == 0 && (2 * S2 * 64) / NT2 >= 1, "the unrolled input prefetch covers the stage input only when the thread count divides it");
  f32x4 xin[XIN];
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NT2;
    int p4 = idx & 63, s = (idx >> 6) & (S2 - 1), t = (idx >> 6) / S2;
    int sb = min(s, B - 1 - b0);
    xin[k] = *goff<const f32x4>(in1 + (size_t)b0 * 512, (uint32_t)(((sb * 2 + t) * 256 + p4 * 4) * 4));
  }
  if (tid < S2) {
    sids[tid] = my_id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)max(my_id, 0) * st::E2_BYTES + st::PHASE);
  }
  load_luts<NT2>(LQ, P.lr_lut, NLR, LA, P.add_lut, NADD);
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NT2;
    int p4 = idx & 63, s = (idx >> 6) & (S2 - 1), t = (idx >> 6) / S2;
    *reinterpret_cast<f32x4*>(&XF[(t * S2 + s) * CS2 + p4 * 4]) = xin[k];
  }
  const auto warm = l2_warm<NT2, 1>(P.warm);
  const auto warm_code = code_warm<NT2>(code_bytes);
  LYRA_SYNC_KEEP();
  TileCtx cx{state, sids, sphase, B - b0, st::E2_BYTES};
  const RbqPre pre1 = resblock_q_prefetch<S2>(cx, 3, st::E_R2_1, P.dwq[0], P.pwq[0], P.cvq[0]);
  // rows read by dependent loads further down the chain (depthwise history of block 0; strided-conv and bottleneck rows)
  static_assert(S2 * ((st::E_R2_0 + 2048 - 1) / 128 + 1) <= NT2 && S2 * 13 <= NT2, "one touch per thread");
  const uint32_t touch0 = state_touch<S2, NT2>(cx, st::E_R2_0, 2 * 256 * 4);
  const uint32_t touch1 = state_touch<S2, NT2>(cx, st::E_D2, 2 * 256 + 2 * 512);

  __syncthreads();   // (the touches above are in flight; XF was written before the ids barrier)

  LYRA_TSTAMP(1);
  // ---- resblock 0, fp32 half: depthwise (dil 1, history 2 rows, replaced) + pointwise 256->256 ----
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int p4 = idx & 63, s = (idx >> 6) & (S2 - 1), t = (idx >> 6) / S2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};   // the chain starts from the bias, which is +0.0 for every depthwise layer (model.hip checks)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int tau = t - (2 - j);
      f32x4 v;
      if (tau >= 0) v = lrelu4(*reinterpret_cast<const f32x4*>(&XF[(tau * S2 + s) * CS2 + p4 * 4]));
      else v = *cx.at<const f32x4>(cx.soff(s) + (uint32_t)(st::E_R2_0 + ((2 + tau) * 256 + p4 * 4) * 4));
      acc = fma4(v, *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(&as_global(P.dw0.w)[j * 256 + p4 * 4]), acc);
    }
    *reinterpret_cast<f32x4*>(&DF[(t * S2 + s) * CS2 + p4 * 4]) = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int p4 = idx & 63, s = (idx >> 6) & (S2 - 1), j = (idx >> 6) / S2;
    if (cx.valid(s))
      *cx.at<f32x4>(cx.soff(s) + (uint32_t)(st::E_R2_0 + (j * 256 + p4 * 4) * 4)) =
          lrelu4(*reinterpret_cast<const f32x4*>(&XF[(j * S2 + s) * CS2 + p4 * 4]));
  }
  LYRA_TSTAMP(2);
  {  // pointwise fp32 -> QUANTIZE -> int8 LeakyReLU -> QP
    f32x4 acc[MT2][2];
    auto aoff = [&](int i, int c) { return (i * 16 + m) * CS2 + c * 16 + q * 4; };
    gemm_f32_bias<MT2, 2, 16, 16>(DF, aoff, P.pw0.w + (wave * 2) * 16 * 64, P.pw0.b, wave * 2 * 16, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 2 + j) * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < MT2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int q8 = quantize_code<MODE>(acc[i][j][e], P.q_r0);
          QP[(i * 16 + q * 4 + e) * QS + n] = (int8_t)lut8(LQ, q8);
        }
    }
  }
  __syncthreads();
  LYRA_TSTAMP(3);
  {  // grouped 1x1 int8 (4 groups 64->64) -> DEQUANTIZE + float skip -> QUANTIZE = X1; operand-swapped (gemm_i8_t)
    static_assert(MT2 == 1, "one 16-row M tile: rows (t, s)");
    const int row = lane & 15, ch0 = wave * 32 + q * 4;
    i32x4 acc[2] = {chan_quad(P.r0b.b, wave * 2), chan_quad(P.r0b.b, wave * 2 + 1)};
    const i32x4 M[2] = {chan_quad(P.r0b.M, wave * 2), chan_quad(P.r0b.M, wave * 2 + 1)};
    const i32x4 sh[2] = {chan_quad(P.r0b.sh, wave * 2), chan_quad(P.r0b.sh, wave * 2 + 1)};
    const int g = wave >> 1;
    auto aoff = [&](int c) { return m * QS + g * 64 + q * 16; };
    gemm_i8_t<2, 1>(QP, aoff, P.r0b.w + (wave * 2) * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = conv_dequant<conv_flavour<MODE, false>()>(acc[j][e], M[j][e], sh[j][e], P.r0b.zout, P.dq_r0.s) +   // dq_r0.z == zout (model.hip checks)
                        XF[row * CS2 + at16(ch0 + 16 * j + e)];
        o[e] = quantize_code<MODE>(v, P.q_x1);
      }
      *reinterpret_cast<int*>(&QX[row * QS + ch0 + 16 * j]) = pack8(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();

  LYRA_TSTAMP(4);
  // ---- int8 resblocks 1, 2 (dilation 3 / 9; ring histories of 6 / 18 rows, T = 2) ---------------
  const RbqPre pre2 = resblock_q_prefetch<S2>(cx, 9, st::E_R2_2, P.dwq[1], P.pwq[1], P.cvq[1]);
  resblock_q256<S2, MODE>(QX, QD, QP, cx, 3, st::E_R2_1, LQ + 1 * 256, LQ + 2 * 256, P.dwq[0], P.pwq[0], P.cvq[0],
                          P.add[0], LA, pre1, 20);
  LYRA_TSTAMP(5);
  resblock_q256<S2, MODE>(QX, QD, QP, cx, 9, st::E_R2_2, LQ + 3 * 256, LQ + 4 * 256, P.dwq[1], P.pwq[1], P.cvq[1],
                          P.add[1], LA + 512, pre2, 30);

  LYRA_TSTAMP(6);
  // ---- int8 LeakyReLU, 2-row history (replaced), conv k4/s2 g4 -> [1][512] --------------------------
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int w4 = idx & 63, s = (idx >> 6) & (S2 - 1), t = (idx >> 6) / S2;
    int w = *reinterpret_cast<const int*>(&QX[(t * S2 + s) * QS + w4 * 4]);
    *reinterpret_cast<int*>(&QB4[((2 + t) * S2 + s) * QS + w4 * 4]) =
        lut8w(LQ + 5 * 256, w);
    *reinterpret_cast<int*>(&QB4[(t * S2 + s) * QS + w4 * 4]) =
        *cx.at<const int>(cx.soff(s) + (uint32_t)(st::E_D2 + t * 256 + w4 * 4));
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int w4 = idx & 63, s = (idx >> 6) & (S2 - 1), t = (idx >> 6) / S2;
    if (cx.valid(s))
      *cx.at<int>(cx.soff(s) + (uint32_t)(st::E_D2 + t * 256 + w4 * 4)) =
          *reinterpret_cast<const int*>(&QB4[((2 + t) * S2 + s) * QS + w4 * 4]);
  }
  LYRA_TSTAMP(7);
#ifdef LYRA_T1_ABL   // TIMING-ONLY ablation (results are wrong): see dec_s0_body
  const bool t1_skip = (tile & 1) != 0;
  i32x4 dacc[1][4] = {};
#else
  constexpr bool t1_skip = false;
  i32x4 dacc[1][4];
#endif
  if (!t1_skip) {  // GEMM rows = streams (rows >= S are over-read padding and discarded)
    const int g = wave >> 1;
    auto aoff = [&](int i, int c) { return (c * S2 + m) * QS + g * 64 + q * 16; };
    gemm_i8<1, 4, 4>(QB4, aoff, P.down2.w + (wave * 4) * 4 * 64, dacc);
  }
  __syncthreads();  // QX/QA/QD/QP and XF are dead from here: QC may overwrite them
  // bottleneck history (ring R=2, T=1): rows [f-2, f-1] -> QC rows 0, 1; new row -> QC row 2
  for (int idx = tid; idx < 2 * S2 * 128; idx += NT2) {
    int w4 = idx & 127, s = (idx >> 7) & (S2 - 1), j = (idx >> 7) / S2;
    int slot = (sphase[s] + j) & 1;
    *reinterpret_cast<int*>(&QC[(j * S2 + s) * QS5 + w4 * 4]) =
        *cx.at<const int>(cx.soff(s) + (uint32_t)(st::E_BOTT + slot * 512 + w4 * 4));
  }
  fold_rows8<2>(dacc[0]);   // lanes 32-63 take over N tiles 2, 3
#pragma unroll
  for (int j = 0; j < (t1_skip ? 0 : 2); ++j) {
    int n = (wave * 4 + j + 2 * (lane >> 5)) * 16 + (lane & 15);
    int bias = as_global(P.down2.b)[n], M = as_global(P.down2.M)[n], sh = as_global(P.down2.sh)[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = (q & 1) * 4 + e;
      int c8 = conv_code<conv_flavour<MODE, false>()>(dacc[0][j][e] + bias, M, sh, P.down2.zout);
      QC[(2 * S2 + s) * QS5 + n] = (int8_t)lut8(LQ + 6 * 256, c8);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < S2 * 128; idx += NT2) {
    int w4 = idx & 127, s = idx >> 7;
    int slot = sphase[s] & 1;
    if (cx.valid(s))
      *cx.at<int>(cx.soff(s) + (uint32_t)(st::E_BOTT + slot * 512 + w4 * 4)) =
          *reinterpret_cast<const int*>(&QC[(2 * S2 + s) * QS5 + w4 * 4]);
  }
  LYRA_TSTAMP(8);
  // ---- bottleneck conv k3 g4: per group K = 3*128, N = 16 -> 64 int8 codes ----------------------------
  if (wave < 4 && !t1_skip) {
    i32x4 acc[1][1];
    const int g = wave;
    auto aoff = [&](int i, int c) { return ((c >> 1) * S2 + m) * QS5 + g * 128 + (c & 1) * 64 + q * 16; };
    gemm_i8<1, 1, 6>(QC, aoff, P.bott.w + g * 6 * 64, acc);
    int n = g * 16 + (lane & 15);
    int bias = as_global(P.bott.b)[n], M = as_global(P.bott.M)[n], sh = as_global(P.bott.sh)[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = q * 4 + e;
      int c8 = conv_code<conv_flavour<MODE, false>()>(acc[0][0][e] + bias, M, sh, P.bott.zout);
      if (s < S2 && cx.valid(s)) {
        *goff<float>(feats + (size_t)b0 * 64, (uint32_t)((s * 64 + n) * 4)) = dequantize_f(c8, P.out);
        if (codes_dbg) codes_dbg[(size_t)(b0 + s) * 64 + n] = (float)c8;
      }
    }
  }
  LYRA_TSTAMP(9);
  LYRA_WSTAMP(101);
  LYRA_WG_END();
  if (tid < S2 && cx.valid(tid)) {
    int ph = sphase[tid] + 1;
    *cx.at<int>(cx.soff(tid) + (uint32_t)(st::PHASE)) = ph >= st::PHASE_MOD ? 0 : ph;
  }
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
  state_touch_sink(touch0 ^ touch1, state, B);
}

}  // namespace lyra
#endif
