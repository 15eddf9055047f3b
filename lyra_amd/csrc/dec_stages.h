// dec_stages.h -- LyraGAN decoder (replaces lyragan.tflite as run by LyraGanModel::RunConditioning /
// RunModel, lyra/lyra_gan_model.cc:53-64, incl. the float->int16 conversion of dsp_utils.h:54-88) as three
// stream-tiled stages (device functions; kernels: dec_kernels.hip one per stage, dec_side_kernel.hip all three).
//
//   dec_s0  8 streams/WG  features -> conv k3 g4 (fp32) -> int8: 4x tconv k4/s2, 3 resblocks @256ch x 2 rows,
//                         2x tconv k4/s2 -> [4][128] fp32
//   dec_s1  8 streams/WG  3 fp32 resblocks @128ch x 4 rows -> tconv k10/s5 (two chained GEMM passes) -> [20][64]
//   dec_s2  4 streams/WG  3 fp32 resblocks @64ch x 20 rows -> tconv k64/s16 -> 320 samples -> int16 PCM
//
// Transposed convs run in polyphase form: output block b (s rows) = bias + [x[b] .. x[b-taps+1]] (K = taps*Cin,
// NEWEST input first = taps ascending) times W[K][s*Cout] -- per output element exactly the oracle's chain, which is
// the order XNNPACK's subconvolution computes.  The tail rows that belong to the next frame are carried in the
// state with the bias removed, as the graph does.
#ifndef LYRA_AMD_CSRC_DEC_STAGES_H_
#define LYRA_AMD_CSRC_DEC_STAGES_H_
#include "resblock_q.h"

namespace lyra {

// =============================================================================================
// stage 0 -- like encoder stage 2 a long chain of small dependent phases: small tile (S = 8 streams, 512 threads,
// ~68 KB LDS), two workgroups per CU.  Rows of [2][S] matrices are t*S + s (one 16-row MFMA tile); GEMMs whose
// rows are just the 8 streams fold two N tiles into the idle upper lanes before their epilogue (fold_rows8).
// =============================================================================================
namespace {
constexpr int SD0 = 8;
constexpr int FS = 72;      // feature row stride (64 + 8) floats
constexpr int CS2 = 264;    // 256 + 8 floats
constexpr int QS = 288;     // int8 row stride, C = 256
constexpr int QS5 = 544;    // int8 row stride, C = 512
constexpr int NTD0 = 512;
constexpr int MTD0 = (2 * SD0) / 16;
constexpr int FB_FLOATS = 3 * 16 * FS;      // sized for a full 16-row M tile (rows >= S are padding)
constexpr int XF_FLOATS = 2 * SD0 * CS2;
constexpr int H8_BYTES = 16 * QS5;
constexpr int QB_BYTES = 2 * SD0 * QS;      // [2][S] rows = one 16-row M tile
constexpr int QA_BYTES = 2 * 16 * QS;       // 2 M tiles of 16 rows: the up1 GEMM reads [t][16 rows]
constexpr int NLR = 6, NADD = 2;            // LeakyReLU / ADD lookup tables (resblock_q.h)
static_assert(SD0 == 8, "tile size the index math below supports");
}  // namespace

__host__ __device__ constexpr size_t dec_s0_lds() {
  return (size_t)(FB_FLOATS + XF_FLOATS) * 4 + H8_BYTES + 3 * QB_BYTES + QA_BYTES + 2 * SD0 * 4 + NLR * 256 +
         NADD * 2048;
}

// feats != nullptr: lossy features [B][64] (GenerativeModel::AddFeatures path).  Otherwise the features are rebuilt
// here from the packets exactly as rvq_decode_kernel does (quantizer.tflite `decode` + DecodeToLossyFeatures,
// residual_vector_quantizer.cc:112-168): ((v0 + v1) + v2) + ... left to right, unused stages contribute v * 0.0f.
// MODE: requantisation flavour, compile-time (see enc_s2_kernel.hip).
template <int MODE>
__device__ __forceinline__ void dec_s0_body(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            float* __restrict__ out0, const uint8_t* __restrict__ packets, int num_stages,
                                            const float* __restrict__ cb, int code_bytes, int tile = (int)blockIdx.x) {
  const DecS0P& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* FB = smem;                                   // [3][16][72]: two history rows + new features (rows s < S used)
  float* XF = FB + FB_FLOATS;                         // [2][S][264]: x164 (float skip of resblock 0)
  int8_t* H8 = reinterpret_cast<int8_t*>(XF + XF_FLOATS);  // [16][544]
  int8_t* QX = H8 + H8_BYTES;
  int8_t* QA = QX + QB_BYTES;
  int8_t* QD = QA + QA_BYTES;
  int8_t* QP = QD + QB_BYTES;
  int* sids = reinterpret_cast<int*>(QP + QB_BYTES);
  int* sphase = sids + SD0;
  int32_t* LA = sphase + SD0;                                // [NADD][2][256] ADD operand tables
  int8_t* LQ = reinterpret_cast<int8_t*>(LA + NADD * 512);   // [NLR][256] LeakyReLU tables
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = tile * SD0;
  wg_schedule_hint();
  LYRA_TSTAMP(80);
  LYRA_WSTAMP(120);
#if !defined(LYRA_WGTRACE_D1) && !defined(LYRA_WGTRACE_D2)
  LYRA_WG_BEGIN();
#endif
  if (tid < SD0) {
    int id = ids[min(b0 + tid, B - 1)];
    sids[tid] = id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)max(id, 0) * st::D0_BYTES + st::PHASE);
  }
  // The packet bytes of this wave's stream (wave = stream, lane = feature channel) depend on nothing but the batch
  // position: requested in the same round trip as the ids / ring phases, not after the barrier that publishes those.
  // Lane i holds byte i; every later use is a v_readlane into an SGPR, so nibble extraction and the codebook row
  // address are scalar work and each of the 46 gathers is one global_load (scalar base + lane offset) and one fma.
  int pkv = 0;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (!feats) {
    const int nbytes = (num_stages + 1) >> 1;
    const uint8_t* pk = packets + (size_t)min(b0 + swave, B - 1) * nbytes;
    if (lane < nbytes) pkv = (int)pk[lane];
  }
  load_luts<NTD0>(LQ, P.lr_lut, NLR, LA, P.add_lut, NADD);
  const auto warm = l2_warm<NTD0, 1>(P.warm);
  const auto warm_code = code_warm<NTD0>(code_bytes);
  const auto warm_cb = l2_warm<NTD0, 1>(feats ? WarmRange{nullptr, 0} : WarmRange{reinterpret_cast<const uint8_t*>(cb), 46 * 16 * 64 * 4});
  LYRA_SYNC_KEEP();
  TileCtx cx{state, sids, sphase, B - b0, st::D0_BYTES};
  // overlap tails / carried rows read by dependent loads further down the chain (state_touch, resblocks.h)
  const uint32_t touch0 = state_touch<SD0, NTD0>(cx, st::D_UP0, 4 * 2 * 64 * 4 + 2 * 256);   // D_UP0 .. D_R0_0
  const uint32_t touch1 = state_touch<SD0, NTD0>(cx, st::D_UP1, 2 * 2 * 64 * 4);

  // ---- feature window [f-2, f-1, f] (history ring R=2, T=1); GEMM rows = streams, 16-row tile ----------
  {
    const int c = tid & 63, s = tid >> 6;          // SD0 * 64 == NTD0: one feature element per thread
    const int b = min(b0 + s, B - 1);
    float f;
    if (feats) {
      f = *goff<const float>(feats + (size_t)b0 * 64, (uint32_t)(((b - b0) * 64 + c) * 4));
    } else {
      f = 0.f;
      // buffer addressing: resource = the codebook, voffset = this lane's channel, soffset = the row (scalar)
      const __amdgpu_buffer_rsrc_t cbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cb), 0, 46 * 16 * 64 * 4, 0x00020000);
#pragma unroll
      for (int k = 0; k < 46; ++k) {   // addresses depend only on the packet: all 46 codebook loads in flight at once
        const int byte = __builtin_amdgcn_readlane(pkv, k >> 1);
        const int used = (k - num_stages) >> 31;                       // all ones / zero, wave-uniform
        const int i = (byte >> ((k & 1) ? 0 : 4)) & 15 & used;
        const float mask = __builtin_bit_cast(float, used & 0x3f800000);   // 1.0f / 0.0f, kept in an SGPR
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cbr, c * 4, (k * 16 + i) * 256, 0));
        // v * mask is exact (v or a signed zero), so fma(v, mask, f) rounds once, exactly like (v * mask) + f
        f = k == 0 ? v * mask : __builtin_fmaf(v, mask, f);
      }
    }
    FB[(2 * 16 + s) * FS + at16(c)] = f;
  }
  for (int idx = tid; idx < 2 * SD0 * 16; idx += NTD0) {
    int p4 = idx & 15, s = (idx >> 4) & (SD0 - 1), j = (idx >> 4) / SD0;
    int slot = (sphase[s] + j) & 1;
    *reinterpret_cast<f32x4*>(&FB[(j * 16 + s) * FS + p4 * 4]) =
        *cx.at<const f32x4>(cx.soff(s) + (uint32_t)(st::D_HEAD + (slot * 64 + p4 * 4) * 4));
  }
  __syncthreads();
  for (int idx = tid; idx < SD0 * 16; idx += NTD0) {
    int p4 = idx & 15, s = idx >> 4;
    int slot = sphase[s] & 1;
    if (cx.valid(s))
      *cx.at<f32x4>(cx.soff(s) + (uint32_t)(st::D_HEAD + (slot * 64 + p4 * 4) * 4)) =
          *reinterpret_cast<const f32x4*>(&FB[(2 * 16 + s) * FS + p4 * 4]);
  }
  LYRA_TSTAMP(81);
#ifdef LYRA_T1_ABL   // TIMING-ONLY ablation (results are wrong): odd tiles skip the GEMM phases of the T = 1 layers -- what a
  const bool t1_skip = (tile & 1) != 0;   // 16-stream tile (twice the streams per 16-row MFMA tile) could return at most
#else
  constexpr bool t1_skip = false;
#endif
  if (!t1_skip) {  // conv k3 g4: per group [16 rows] x K=48 x N=128; LeakyReLU; QUANTIZE -> H8
    f32x4 acc[1][4];
    const int g = wave >> 1;
    auto aoff = [&](int i, int c) { return (c * 16 + m) * FS + g * 16 + q * 4; };
    gemm_f32_bias<1, 4, 3, 3>(FB, aoff, P.head.w + (wave * 4) * 3 * 64, P.head.b, wave * 4 * 16, acc);
    fold_rows8<2>(acc[0]);   // rows = 8 streams: lanes 32-63 take over N tiles 2, 3
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 4 + j + 2 * (lane >> 5)) * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        H8[((q & 1) * 4 + e) * QS5 + n] = (int8_t)quantize_code<MODE>(lrelu(acc[0][j][e]), P.q0);
    }
  }
  __syncthreads();
  LYRA_TSTAMP(82);
  if (!t1_skip) {  // 4 grouped int8 transposed convs k4/s2 (one input row -> 4 output rows), carried tail of 2 rows
    i32x4 acc[1][8];
    const int g = wave >> 1;
    const TconvQ U = P.up0[g];
    auto aoff = [&](int i, int c) { return m * QS5 + g * 128 + c * 64 + q * 16; };
    // after fold_rows8 lane L works on channel tile ct, streams (q & 1) * 4 + e: request its epilogue operands
    // (carried tail rows, bias, fold terms) before the GEMM so that their latency overlaps it
    const int ct = (wave & 1) * 2 + (lane >> 5);
    const int co = ct * 16 + (lane & 15);
    const int bias = as_global(U.bias)[co];
    const float sub = as_global(P.up0_sub[g])[co];
    const int pc = at16(g * 64 + co);
    int zfv[4];
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) zfv[tap] = as_global(U.zfold)[tap * 64 + co];
    float told[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t stp = cx.soff((q & 1) * 4 + e) + (uint32_t)(st::D_UP0 + g * 512 + co * 4);
      told[0][e] = *cx.at<const float>(stp);
      told[1][e] = *cx.at<const float>(stp + 256);
    }
    gemm_i8<1, 8, 2>(H8, aoff, U.w + ((wave & 1) * 8) * 2 * 64, acc);
    fold_rows8<4>(acc[0]);   // lanes 32-63 take over this wave's second channel tile (N tiles 4..7)
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int zf = zfv[tap];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = (q & 1) * 4 + e;
        const uint32_t stp = cx.soff(s) + (uint32_t)(st::D_UP0 + g * 512 + co * 4);
        float y = conv_dequant<conv_flavour<MODE, false>()>(acc[0][tap][e] + zf + bias, U.M, U.sh, U.zout, P.up0_dq[g].s);   // up0_dq.z == zout (model.hip)
        if (tap < 2) {
          y = y + told[tap < 2 ? tap : 0][e];
          XF[(tap * SD0 + s) * CS2 + pc] = y;
        } else {
          y = y + 0.f;
          if (cx.valid(s)) *cx.at<float>(stp + (uint32_t)((tap - 2) * 256)) = y - sub;
        }
      }
    }
  }
  __syncthreads();
  LYRA_TSTAMP(83);
  // ---- a0 = QUANTIZE(lrelu(x164)); resblock 0 (int8 body, float skip) depthwise, dilation 1 --------------
  // Thread (s, w4) owns channels 4*w4 .. 4*w4+3 of stream s for both rows: quantize -> depthwise over
  // [a(t-2), a(t-1), a(t)] -> history (2 rows, replaced) stay in registers; no LDS round trip, one barrier.
  const RbqPre pre1 = resblock_q_prefetch<SD0>(cx, 3, st::D_R0_1, P.dwq[1], P.pwq[1], P.cvq[1]);
  {
    const DwQ dq = P.dwq[0];
    const int w4 = tid & 63, s = tid >> 6;
    const uint32_t hp = cx.soff(s) + (uint32_t)(st::D_R0_0 + w4 * 4);
    const int h0 = *cx.at<const int>(hp), h1 = *cx.at<const int>(hp + 256);
    int ww[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) ww[j] = *reinterpret_cast<const int LYRA_GLOBAL*>(&as_global(dq.w)[j * 256 + w4 * 4]);
    const i32x4 db = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.b)[w4 * 4]);
    const i32x4 dM = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.M)[w4 * 4]);
    const i32x4 dsh = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.sh)[w4 * 4]);
    int a[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int c8[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        c8[e] = quantize_code<MODE>(lrelu(XF[(t * SD0 + s) * CS2 + at16(w4 * 4 + e)]), P.q1);
      a[t] = pack8(c8[0], c8[1], c8[2], c8[3]);
    }
    const int x[2][3] = {{h0, h1, a[0]}, {h1, a[0], a[1]}};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int acc = db[e];                                  // zero point folded into dq.b
#pragma unroll
        for (int j = 0; j < 3; ++j) acc += sx8(x[t][j], e) * sx8(ww[j], e);
        o[e] = conv_code<conv_flavour<MODE, false>()>(acc, dM[e], dsh[e], dq.zout);
      }
      *reinterpret_cast<int*>(&QD[(t * SD0 + s) * QS + w4 * 4]) = pack8(o[0], o[1], o[2], o[3]);
      if (cx.valid(s)) *cx.at<int>(hp + (uint32_t)(t * 256)) = a[t];
    }
    __syncthreads();
    LYRA_TSTAMP(84);
    static_assert(MTD0 == 1, "one 16-row M tile: rows (t, s)");
    const int row = lane & 15, ch0 = wave * 32 + q * 4;   // operand-swapped GEMMs (lyra_dev.h gemm_i8_t): this lane's row / channels
    {
      i32x4 acc[2] = {chan_quad(P.pwq[0].b, wave * 2), chan_quad(P.pwq[0].b, wave * 2 + 1)};
      const i32x4 M[2] = {chan_quad(P.pwq[0].M, wave * 2), chan_quad(P.pwq[0].M, wave * 2 + 1)};
      const i32x4 sh[2] = {chan_quad(P.pwq[0].sh, wave * 2), chan_quad(P.pwq[0].sh, wave * 2 + 1)};
      auto aoff = [&](int c) { return m * QS + c * 64 + q * 16; };
      gemm_i8_t<2, 4>(QD, aoff, P.pwq[0].w + (wave * 2) * 4 * 64, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r8[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r8[e] = lut8(LQ, conv_code<conv_flavour<MODE, true>()>(acc[j][e], M[j][e], sh[j][e], P.pwq[0].zout));   // pointwise: ungrouped
        *reinterpret_cast<int*>(&QP[row * QS + ch0 + 16 * j]) = pack8(r8[0], r8[1], r8[2], r8[3]);
      }
    }
    __syncthreads();
    {
      i32x4 acc[2] = {chan_quad(P.cvq[0].b, wave * 2), chan_quad(P.cvq[0].b, wave * 2 + 1)};
      const i32x4 M[2] = {chan_quad(P.cvq[0].M, wave * 2), chan_quad(P.cvq[0].M, wave * 2 + 1)};
      const i32x4 sh[2] = {chan_quad(P.cvq[0].sh, wave * 2), chan_quad(P.cvq[0].sh, wave * 2 + 1)};
      const int g = wave >> 1;
      auto aoff = [&](int c) { return m * QS + g * 64 + q * 16; };
      gemm_i8_t<2, 1>(QP, aoff, P.cvq[0].w + (wave * 2) * 64, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = conv_dequant<conv_flavour<MODE, false>()>(acc[j][e], M[j][e], sh[j][e], P.cvq[0].zout, P.dq_r0.s) +   // dq_r0.z == zout
                          XF[row * CS2 + at16(ch0 + 16 * j + e)];
          o[e] = quantize_code<MODE>(v, P.q3);
        }
        *reinterpret_cast<int*>(&QX[row * QS + ch0 + 16 * j]) = pack8(o[0], o[1], o[2], o[3]);
      }
    }
    __syncthreads();
  }
  LYRA_TSTAMP(85);
  const RbqPre pre2 = resblock_q_prefetch<SD0>(cx, 9, st::D_R0_2, P.dwq[2], P.pwq[2], P.cvq[2]);
  resblock_q256<SD0, MODE>(QX, QD, QP, cx, 3, st::D_R0_1, LQ + 1 * 256, LQ + 2 * 256, P.dwq[1], P.pwq[1], P.cvq[1],
                           P.add[0], LA, pre1, 90);
  resblock_q256<SD0, MODE>(QX, QD, QP, cx, 9, st::D_R0_2, LQ + 3 * 256, LQ + 4 * 256, P.dwq[2], P.pwq[2], P.cvq[2],
                           P.add[1], LA + 512, pre2, 94);
  LYRA_TSTAMP(86);
  // a = int8 LeakyReLU(X3), laid out for the up1 GEMM as [t][16 rows][QS] (rows s >= S are padding)
  for (int idx = tid; idx < 2 * SD0 * 64; idx += NTD0) {
    int w4 = idx & 63, s = (idx >> 6) & (SD0 - 1), t = (idx >> 6) / SD0;
    int w = *reinterpret_cast<const int*>(&QX[(t * SD0 + s) * QS + w4 * 4]);
    *reinterpret_cast<int*>(&QA[(t * 16 + s) * QS + w4 * 4]) =
        lut8w(LQ + 5 * 256, w);
  }
  __syncthreads();
  LYRA_TSTAMP(87);
  {  // 2 grouped int8 transposed convs k4/s2: rows t=0,1 -> 6 output rows (integer overlap-add), tail of 2
    i32x4 acc[2][4];
    const int g = wave >> 2, ct = wave & 3;
    const TconvQ U = P.up1[g];
    auto aoff = [&](int i, int c) { return (i * 16 + m) * QS + g * 128 + c * 64 + q * 16; };
    gemm_i8<2, 4, 2>(QA, aoff, U.w + (ct * 4) * 2 * 64, acc);
    const int co = ct * 16 + (lane & 15);
    const int bias = as_global(U.bias)[co];
    const float sub = as_global(P.up1_sub[g])[co];
    const int pc = at16(g * 64 + co);
    int zf[4];
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) zf[tap] = as_global(U.zfold)[tap * 64 + co];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = q * 4 + e;
      if (s >= SD0) continue;
      const uint32_t stp = cx.soff(s) + (uint32_t)(st::D_UP1 + g * 512 + co * 4);
      int o[6];
      o[0] = acc[0][0][e] + zf[0];
      o[1] = acc[0][1][e] + zf[1];
      o[2] = (acc[0][2][e] + zf[2]) + (acc[1][0][e] + zf[0]);
      o[3] = (acc[0][3][e] + zf[3]) + (acc[1][1][e] + zf[1]);
      o[4] = acc[1][2][e] + zf[2];
      o[5] = acc[1][3][e] + zf[3];
      float y[6];
#pragma unroll
      for (int tau = 0; tau < 6; ++tau) {
        y[tau] = conv_dequant<conv_flavour<MODE, false>()>(o[tau] + bias, U.M, U.sh, U.zout, P.up1_dq[g].s);   // up1_dq.z == zout (model.hip)
      }
      y[0] = y[0] + *cx.at<const float>(stp);
      y[1] = y[1] + *cx.at<const float>(stp + 256);
#pragma unroll
      for (int tau = 2; tau < 6; ++tau) y[tau] = y[tau] + 0.f;
      if (cx.valid(s)) {
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) *goff<float>(out0 + (size_t)b0 * 512, (uint32_t)(((s * 4 + tau) * 128 + pc) * 4)) = y[tau];
        *cx.at<float>(stp) = y[4] - sub;
        *cx.at<float>(stp + 256) = y[5] - sub;
      }
    }
  }
  LYRA_TSTAMP(88);
  LYRA_WSTAMP(121);
#if !defined(LYRA_WGTRACE_D1) && !defined(LYRA_WGTRACE_D2)
  LYRA_WG_END();
#endif
  if (tid < SD0 && cx.valid(tid)) {   // this region's ring phase
    int ph = sphase[tid] + 1;
    *cx.at<int>(cx.soff(tid) + (uint32_t)(st::PHASE)) = ph >= st::PHASE_MOD ? 0 : ph;
  }
  l2_warm_sink(warm_cb, state, B);
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
  state_touch_sink(touch0 ^ touch1, state, B);
}

// =============================================================================================
// stage 1
// =============================================================================================
namespace {
constexpr int SD1 = 8;
constexpr int CS1 = 136;
#ifndef LYRA_S1_THREADS
#define LYRA_S1_THREADS 512   // 8 waves per tile: 4 waves per SIMD with two tiles per CU (256 = the 4-wave layout)
#endif
constexpr int NTD1 = LYRA_S1_THREADS;
}  // namespace

// tconv k10/s5, polyphase: output block b (5 rows x 64 ch = N 320) = bias, then x[b] . W[taps 0..4], then
// x[b-1] . W[taps 5..9]: ONE fp32 chain per output, taps ascending = NEWEST input first (XNNPACK's subconvolution
// order, tests/test_xnnpack_witness.py).  Pass 1 runs every input row t against taps 0..4 from the bias (the head of
// block t's chain; block 0 is complete after it, its older input being the previous frame's carried tail); the C
// tiles are then shifted UP by one input row (8 of a tile's 16 rows: a 32-lane rotation) so that C row (b, s) sits on
// A row (b - 1, s), block 4 enters as the bare bias, and pass 2 continues the chains with taps 5..9.
// Block 4 = bias + x[3] . W[5..9] is the carried tail.  A wave computes NTW of the 20 N tiles starting at tile0.
// HB: LDS scratch of this wave (NTW x 4 x 64 floats): block 0 -- complete after pass 1 -- waits there for the epilogue
// instead of in 4 NTW registers across pass 2 (at the 128-VGPR cap they spilled).
template <int NTW>
__device__ __forceinline__ void dec_s1_tconv(const float* XB, const float* SB, const DecS1P& P, const TileCtx& cx,
                                             int b0, float* __restrict__ out1, int tile0, float* HB) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  f32x4 acc[2][NTW];
  auto aoff = [&](int i, int c) {
    int R = i * 16 + m, t = R / SD1, s = R & (SD1 - 1);
    return (t * SD1 + s) * CS1 + c * 16 + q * 4;
  };
  const f32x4* wfrag = P.up.w + tile0 * 16 * 64;   // per N tile 16 K chunks: 0-7 taps 0..4, 8-15 taps 5..9
  {
    f32x4 bias4[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float v = as_global(P.up.b)[((tile0 + j) * 16 + (lane & 15)) & 63];
      bias4[j] = (f32x4){v, v, v, v};
    }
    gemm_f32_init<2, NTW, 8, 16>(XB, aoff, wfrag, bias4, acc);
  }
  LYRA_TSTAMP(54);
  LYRA_WSTAMP(114);
  const bool lo = lane < 32;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const float biasb = as_global(P.up.b)[((tile0 + j) * 16 + (lane & 15)) & 63];   // L1 hit: read before pass 1 too
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = acc[0][j][e], bb = acc[1][j][e];   // a = [Y(b0) | Y(b1)], bb = [Y(b2) | Y(b3)]  (lanes 0-31 | 32-63)
      HB[(j * 4 + e) * 64 + lane] = a;             // lanes 0-31: block 0, complete (parked in LDS, read back by the same lane)
      rot32_pair(a, bb);                           // a = [Y(b1) | Y(b0)], bb = [Y(b3) | Y(b2)]
      acc[0][j][e] = lo ? a : bb;                  // blocks 1 | 2
      acc[1][j][e] = lo ? bb : biasb;              // blocks 3 | 4 (block 4 starts from the bare bias)
    }
  }
  gemm_f32<2, NTW, 8, 16, false>(XB, aoff, wfrag + 8 * 64, acc);
  LYRA_TSTAMP(55);
  LYRA_WSTAMP(115);
  int lane_e = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_e));
  const int q_e = lane_e >> 4;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int n = (tile0 + j) * 16 + (lane_e & 15);
    const int jj = n >> 6, co = n & 63;
    const float sub = as_global(P.up_sub)[co];
    const int pc = at16(co);
    // C rows (SD1 = 8): tile 0 = blocks 1 | 2, tile 1 = blocks 3 | 4 (lanes 0-31 | 32-63); block 4 is the carried tail
    const bool lo = lane_e < 32;
    const int blk = lo ? 0 : 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = (q_e & 1) * 4 + e;
      const float y0 = acc[0][j][e] + 0.f, y1 = acc[1][j][e] + 0.f;
      if (!cx.valid(s)) continue;
      float LYRA_GLOBAL* o = goff<float>(out1 + (size_t)b0 * 1280, (uint32_t)(((s * 20 + 5 * (1 + blk) + jj) * 64 + pc) * 4));
      o[0] = y0;                                   // block 1 | 2
      if (lo) o[10 * 64] = y1;                     // block 3
      else *cx.at<float>(cx.soff(s) + (uint32_t)(st::D_UP2 + (jj * 64 + co) * 4)) = y1 - sub;
    }
    if (lo) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = q_e * 4 + e;
        const float y = HB[(j * 4 + e) * 64 + lane_e] + SB[(jj * SD1 + s) * 72 + co];
        if (cx.valid(s)) *goff<float>(out1 + (size_t)b0 * 1280, (uint32_t)(((s * 20 + jj) * 64 + pc) * 4)) = y;
      }
    }
  }
}

__host__ __device__ constexpr size_t dec_s1_lds() { return (size_t)(4 * SD1 * CS1 + 4 * SD1 * CS1 + 4 * SD1 * CS1 + 5 * SD1 * 72) * 4 + 2 * SD1 * 4; }

__device__ __forceinline__ void dec_s1_body(const DecS1P& P, const float* __restrict__ in0,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            float* __restrict__ out1, int code_bytes, int tile = (int)blockIdx.x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                     // [4][S][136]: X[t]
  float* DB = XB + 4 * SD1 * CS1;       // [4][S][136]
  float* PB = DB + 4 * SD1 * CS1;       // [4][S][136]
  float* SB = PB + 4 * SD1 * CS1;       // [5][S][72]: tail of the previous frame's transposed conv
  int* sids = reinterpret_cast<int*>(SB + 5 * SD1 * 72);
  int* sphase = sids + SD1;
  wg_schedule_hint();
  const int tid = threadIdx.x, wave = tid >> 6;
  const int b0 = tile * SD1;
  LYRA_TSTAMP(50);
#ifdef LYRA_WGTRACE_D1   // per-workgroup trace of THIS kernel instead of dec_s0 (wg_trace_full.py (a probe of an earlier round, removed since: git history))
  LYRA_WG_BEGIN();
#endif
  // the stage input does not depend on the stream ids: requested with them, ahead of the barrier (see enc_s1_body)
  int my_id = 0;
  if (tid < SD1) my_id = ids[min(b0 + tid, B - 1)];
  constexpr int XIN = (4 * SD1 * 32) / NTD1;
  static_assert((4 * SD1 * 32) % NTD1 == 0 && (4 * SD1 * 32) / NTD1 >= 1, "the unrolled input prefetch covers the stage input only when the thread count divides it");
  f32x4 xin[XIN];
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NTD1;
    int p4 = idx & 31, s = (idx >> 5) & (SD1 - 1), t = (idx >> 5) / SD1;
    int sb = min(s, B - 1 - b0);
    xin[k] = *goff<const f32x4>(in0 + (size_t)b0 * 512, (uint32_t)(((sb * 4 + t) * 128 + p4 * 4) * 4));
  }
  if (tid < SD1) {
    sids[tid] = my_id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)max(my_id, 0) * st::D1_BYTES + st::PHASE);
  }
  const auto warm = l2_warm<NTD1, 2>(P.warm);
  const auto warm_code = code_warm<NTD1>(code_bytes);
#pragma unroll
  for (int k = 0; k < XIN; ++k) {
    const int idx = tid + k * NTD1;
    int p4 = idx & 31, s = (idx >> 5) & (SD1 - 1), t = (idx >> 5) / SD1;
    *reinterpret_cast<f32x4*>(&XB[(t * SD1 + s) * CS1 + p4 * 4]) = xin[k];
  }
  LYRA_SYNC_KEEP();
  TileCtx cx{state, sids, sphase, B - b0, st::D1_BYTES};
  const auto H0 = hist128_prefetch<SD1, NTD1>(cx, 1, st::D_R1_0);   // first block's history: same round trip as the input
  for (int idx = tid; idx < 5 * SD1 * 16; idx += NTD1) {   // carried tail: used at the end
    int p4 = idx & 15, s = (idx >> 4) & (SD1 - 1), j = (idx >> 4) / SD1;
    *reinterpret_cast<f32x4*>(&SB[(j * SD1 + s) * 72 + p4 * 4]) =
        *cx.at<const f32x4>(cx.soff(s) + (uint32_t)(st::D_UP2 + (j * 64 + p4 * 4) * 4));
  }
  __syncthreads();
  LYRA_TSTAMP(51);
  LYRA_WSTAMP(111);
  resblocks128<SD1, NTD1>(XB, DB, PB, cx, P.dw, P.pw, P.cv, st::D_R1_0, st::D_R1_1, st::D_R1_2, H0);
  LYRA_TSTAMP(52);
  LYRA_WSTAMP(112);
  for (int idx = tid; idx < 4 * SD1 * 32; idx += NTD1) {
    int p4 = idx & 31, rs = idx >> 5;
    f32x4* x = reinterpret_cast<f32x4*>(&XB[rs * CS1 + p4 * 4]);
    *x = lrelu4(*x);
  }
  __syncthreads();
  LYRA_TSTAMP(53);
  LYRA_WSTAMP(113);
  // transposed conv k10/s5: 20 N tiles over the waves (5 each with 4 waves; 3,3,3,3,2,2,2,2 with 8)
  // (DB and PB, 2 x 17 KB, are free once the residual blocks are done: the waves' head scratch, 20 KB in all, fits)
  constexpr int HEAD_TILES = NTD1 == 256 ? 5 : 3;
  static_assert((NTD1 / 64) * HEAD_TILES * 256 <= 2 * 4 * SD1 * CS1, "head scratch fits into DB + PB");
  float* HB = DB + wave * (HEAD_TILES * 4 * 64);
  if (NTD1 == 256) dec_s1_tconv<5>(XB, SB, P, cx, b0, out1, wave * 5, HB);
  else if (wave < 4) dec_s1_tconv<3>(XB, SB, P, cx, b0, out1, wave * 3, HB);
  else dec_s1_tconv<2>(XB, SB, P, cx, b0, out1, 12 + (wave - 4) * 2, HB);
#ifdef LYRA_WGTRACE_D1
  __syncthreads();
  LYRA_WG_END();
#endif
  if (tid < SD1 && cx.valid(tid)) {   // this region's ring phase
    int ph = sphase[tid] + 1;
    *cx.at<int>(cx.soff(tid) + (uint32_t)(st::PHASE)) = ph >= st::PHASE_MOD ? 0 : ph;
  }
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
}

// =============================================================================================
// stage 2: SD2 streams per workgroup of 64 * SD2 threads (4 / 256 or 8 / 512)
// =============================================================================================
namespace {
constexpr int CS0 = 72;
}  // namespace
__host__ __device__ constexpr size_t dec_s2_lds(int sd2) { return (size_t)(27 * sd2 * CS0 + sd2 * 48) * 4 + 64; }

template <int SD2>
__device__ __forceinline__ void dec_s2_body(const DecS2P& P, const float* __restrict__ in1,
                                            const int32_t* __restrict__ ids, int B, uint8_t* __restrict__ state,
                                            int16_t* __restrict__ pcm, int code_bytes, int tile = (int)blockIdx.x) {
  constexpr int NTD2 = 64 * SD2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                     // [27][S][72]: rows 0-2 zeros, rows 3-22 activations, rows 23-26 zeros
  float* SB = XB + 27 * SD2 * CS0;      // old overlap tail [S][48]
  int* sids = reinterpret_cast<int*>(SB + SD2 * 48);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  wg_schedule_hint();
  const int b0 = tile * SD2;
  LYRA_TSTAMP(60);
#ifdef LYRA_WGTRACE_D2
  LYRA_WG_BEGIN();
#endif
  int my_id = 0;
  if (tid < SD2) my_id = ids[min(b0 + tid, B - 1)];
  const int wn = wave & 3, wm = wave >> 2;
  const int pcol = at16(wn * 16 + (lane & 15));
  // the stage input does not depend on the stream ids: requested with them, ahead of the barrier (see enc_s1_body)
  constexpr bool SW = LYRA_SWAP64 != 0;   // operand-swapped GEMMs: a lane holds 4 consecutive channels of ONE row (lyra_dev.h)
  f32x4 xr[5][1];  // residual stream in registers (MFMA C layout; transposed with SW: see resblocks64r)
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if constexpr (SW) {
      const int R = (wm * 5 + i) * 16 + m, t = R / SD2, s = R & (SD2 - 1);
      const int sb = min(s, B - 1 - b0);
      xr[i][0] = *goff<const f32x4>(in1 + (size_t)b0 * 1280, (uint32_t)(((sb * 20 + t) * 64 + wn * 16 + q * 4) * 4));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int R = (wm * 5 + i) * 16 + q * 4 + e, t = R / SD2, s = R & (SD2 - 1);
        int sb = min(s, B - 1 - b0);
        xr[i][0][e] = *goff<const float>(in1 + (size_t)b0 * 1280, (uint32_t)(((sb * 20 + t) * 64 + pcol) * 4));
      }
    }
  }
  if (tid < SD2) sids[tid] = my_id;
  const auto warm = l2_warm<NTD2, 1>(P.warm);
  const auto warm_code = code_warm<NTD2>(code_bytes);
  LYRA_SYNC_KEEP();
  TileCtx cx{state, sids, nullptr, B - b0, st::D2_BYTES};   // T = 20 >= every 2*dilation: no ring, no phase
  for (int idx = tid; idx < 7 * SD2 * 16; idx += NTD2) {
    int p4 = idx & 15, s = (idx >> 4) & (SD2 - 1), j = (idx >> 4) / SD2;
    int row = j < 3 ? j : 20 + j;
    *reinterpret_cast<f32x4*>(&XB[(row * SD2 + s) * CS0 + p4 * 4]) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int idx = tid; idx < SD2 * 48; idx += NTD2) {
    int s = idx / 48, i = idx - s * 48;
    SB[idx] = *cx.at<const float>(cx.soff(s) + (uint32_t)(st::D_UP3 + i * 4));
  }
  LYRA_TSTAMP(61);
  resblocks64r<SD2, NTD2>(xr, XB + 3 * SD2 * CS0, cx, P.dw, P.pw, P.cv, st::D_R2_0, st::D_R2_1, st::D_R2_2);
  LYRA_TSTAMP(62);
#pragma unroll
  for (int i = 0; i < 5; ++i)
{
    const f32x4 a4 = lrelu4(xr[i][0]);
    if constexpr (SW) {
      *reinterpret_cast<f32x4*>(&XB[(3 * SD2 + (wm * 5 + i) * 16 + m) * CS0 + wn * 16 + q * 4]) = a4;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) XB[(3 * SD2 + (wm * 5 + i) * 16 + q * 4 + e) * CS0 + pcol] = a4[e];
    }
  }
  __syncthreads();
  // tconv k64/s16, polyphase: blocks b = 0..22 (+1 of padding), rows (b, s); K = 4 x 64 (newest input first: chunk
  // group g reads input block b - g = LDS row b + 3 - g); N = 16 phases; every chain starts from the bias.
  // 24*S rows = 6 M tiles per 4 streams: waves 0..(3 * S / 4 - 1) take two each.
  if (wave < 3 * SD2 / 4) {
    f32x4 acc[2][1];
    auto aoff = [&](int i, int c) {
      int R = (2 * wave + i) * 16 + m, b = R / SD2, s = R & (SD2 - 1);
      return ((b + 3 - (c >> 2)) * SD2 + s) * CS0 + (c & 3) * 16 + q * 4;
    };
    const float bias = as_global(P.up.b)[0];
    const f32x4 bias4[1] = {(f32x4){bias, bias, bias, bias}};
    gemm_f32_init<2, 1, 16, 16, gemm_pf<2, 1>(), SW>(XB, aoff, P.up.w, bias4, acc);   // (one output channel: the splat is every phase's bias)
    const int j = lane & 15;
    if constexpr (SW) {
      // row (b, s) = this lane's C column; its four values are phases 4q .. 4q + 3 = four CONSECUTIVE output samples
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int R = (2 * wave + i) * 16 + m, b = R / SD2, s = R & (SD2 - 1);
        if (b > 22) continue;
        const int tau0 = 16 * b + 4 * q;
        f32x4 y = acc[i][0];
        f32x4 old = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (tau0 < 48) old = *reinterpret_cast<const f32x4*>(&SB[s * 48 + tau0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = y[e] + old[e];   // (+ 0.f beyond the old tail, as the graph's ADD does)
        if (!cx.valid(s)) continue;
        if (tau0 < 320) {
          // UnitToInt16Scalar (dsp_utils.h:54-88): scale, clip, C truncation
          int v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f = y[e] * 32768.f;
            f = f < -32768.f ? -32768.f : f;
            f = f > 32767.f ? 32767.f : f;
            v[e] = (int)(int16_t)f;
          }
          typedef int i32x2 __attribute__((ext_vector_type(2)));
          const i32x2 pk = (i32x2){(v[0] & 0xffff) | (v[1] << 16), (v[2] & 0xffff) | (v[3] << 16)};
          *goff<i32x2>(pcm + (size_t)b0 * 320, (uint32_t)((s * 320 + tau0) * 2)) = pk;
        } else {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = y[e] - P.up_sub;
          *cx.at<f32x4>(cx.soff(s) + (uint32_t)(st::D_UP3 + (tau0 - 320) * 4)) = t;
        }
      }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int R = (2 * wave + i) * 16 + q * 4 + e, b = R / SD2, s = R & (SD2 - 1);
        if (b > 22) continue;
        const int tau = 16 * b + j;
        float y = acc[i][0][e];
        y = y + (tau < 48 ? SB[s * 48 + tau] : 0.f);
        if (!cx.valid(s)) continue;
        if (tau < 320) {
          // UnitToInt16Scalar (dsp_utils.h:54-88): scale, clip, C truncation
          float v = y * 32768.f;
          v = v < -32768.f ? -32768.f : v;
          v = v > 32767.f ? 32767.f : v;
          *goff<int16_t>(pcm + (size_t)b0 * 320, (uint32_t)((s * 320 + tau) * 2)) = (int16_t)v;
        } else {
          *cx.at<float>(cx.soff(s) + (uint32_t)(st::D_UP3 + (tau - 320) * 4)) = y - P.up_sub;
        }
      }
  }
#ifdef LYRA_WGTRACE_D2
  __syncthreads();
  LYRA_WG_END();
#endif
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
}

}  // namespace lyra
#endif
