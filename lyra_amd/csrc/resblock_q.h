// resblock_q.h -- the int8 residual block at 256 channels x 2 rows shared by encoder stage 2 and decoder
// stage 0 (graph ops 107-133 of soundstream_encoder.tflite / 93-117 of lyragan.tflite), plus the lookup
// tables that replace the int8 LeakyReLU / ADD rescaling arithmetic.
//
// These stages are bound by VALU issue (fixed-point emulation) and by exposed global-load latency, not by
// MFMA or HBM, so:
//   * int8 LeakyReLU is a function of one int8 code -> a 256-byte table in LDS (built on the host with the
//     mode's arithmetic -- gemmlowp or XNNPACK's Q8 multipliers --, model.hip lrelu_luts); the TFLite-builtin int8
//     ADD's two operand rescalings are int32 tables (XNNPACK's ADD is cheap enough to compute).
//   * thread (s, w4) owns channels 4*w4..4*w4+3 of stream s for BOTH rows of the frame, so LeakyReLU ->
//     depthwise conv -> history write are thread-local (no LDS round trip, no barrier);
//   * every global load a block needs (ring history words, per-channel requantisation parameters) is issued
//     by resblock_q_prefetch() ahead of the previous phase's work.
// QX residual stream, QD/QP scratch, all [2][S][288] int8.  Ring history of R2 = 2*d rows, T = 2.
#pragma once
#include "resblocks.h"

namespace lyra {

__device__ __forceinline__ int sx8(int w, int i) { return (int)(int8_t)(w >> (8 * i)); }
// the low bytes of four registers -> one dword: two v_perm_b32 (selector 0-3: bytes of the second operand, 4-7: of the
// first, 0x0c: zero) and an or
__device__ __forceinline__ int pack8(int a, int b, int c, int d) {
  const unsigned lo = __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x0c0c0400u);
  const unsigned hi = __builtin_amdgcn_perm((unsigned)d, (unsigned)c, 0x04000c0cu);
  return (int)(lo | hi);
}

// byte e of three registers -> (x0[e], x1[e], x2[e], 0)
__device__ __forceinline__ int taps3(int x0, int x1, int x2, int e) {
  const unsigned t01 = __builtin_amdgcn_perm((unsigned)x1, (unsigned)x0, 0x0c0c0000u | ((4u + e) << 8) | (unsigned)e);
  return (int)__builtin_amdgcn_perm((unsigned)x2, t01, 0x0c000100u | ((4u + e) << 16));
}

// ---- lookup tables (LDS) ----------------------------------------------------------------------------
// -DLYRA_LUT_IDENTITY: TIMING-ONLY ablation (results are wrong) -- the table reads vanish; what the step gains is ALL the
// byte-table gathers can cost (round 6, profiles/r06_ab_lut_identity.txt).
#ifdef LYRA_LUT_IDENTITY
__device__ __forceinline__ int lut8(const int8_t*, int c8) { return c8; }
#else
__device__ __forceinline__ int lut8(const int8_t* lut, int c8) { return (int)lut[c8 + 128]; }
#endif
// four packed codes at once
__device__ __forceinline__ int lut8w(const int8_t* lut, int w) {
#ifdef LYRA_LUT_IDENTITY
  return w;
#endif
  const uint32_t u = (uint32_t)w ^ 0x80808080u;
  return pack8(lut[u & 255], lut[(u >> 8) & 255], lut[(u >> 16) & 255], lut[u >> 24]);
}
__device__ __forceinline__ int add_q_lut(const int32_t* lut, int a, int b, const AddQ& L) {
  return clamp8(mbqm_double(lut[a + 128] + lut[256 + b + 128], L.mo, L.so) + L.zo);
}
// copies n_lr LeakyReLU tables and n_add ADD tables from the weight arena to LDS (caller syncs)
template <int NT>
__device__ __forceinline__ void load_luts(int8_t* LQ, const int8_t* lr_lut, int n_lr, int32_t* LA,
                                          const int32_t* add_lut, int n_add) {
  for (int i = threadIdx.x; i < n_lr * 64; i += NT)
    reinterpret_cast<int*>(LQ)[i] = as_global(reinterpret_cast<const int*>(lr_lut))[i];
  for (int i = threadIdx.x; i < n_add * 512; i += NT) LA[i] = as_global(add_lut)[i];
}

// ---- per-thread prefetch of everything global a block touches -------------------------------------------
struct RbqPre {
  int h[2][2];        // ring history words: [t][0] = row t - 2d, [t][1] = row t - d
  int ww[3];          // depthwise taps, 4 channels each
  i32x4 b, M, sh;     // depthwise requantisation, 4 channels
};

template <int S>
__device__ __forceinline__ RbqPre resblock_q_prefetch(const TileCtx& cx, int d, int off, const DwQ& dq,
                                                      const ConvQ& pw, const ConvQ& cv) {
  static_assert(S == 8, "thread <-> (stream, channel word) mapping below assumes 8 streams x 64 words = 512 threads");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w4 = tid & 63, s = tid >> 6;
  const int R2 = 2 * d;
  const int base = (cx.sphase[s] * 2) % R2;
  RbqPre p;
  const uint32_t hp = cx.soff(s) + (uint32_t)(off + w4 * 4);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int r1 = base + t + d;
    r1 = r1 >= R2 ? r1 - R2 : r1;
    p.h[t][0] = *cx.at<const int>(hp + (uint32_t)((base + t) * 256));   // row t - 2d (about to be replaced)
    p.h[t][1] = *cx.at<const int>(hp + (uint32_t)(r1 * 256));
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) p.ww[j] = *reinterpret_cast<const int LYRA_GLOBAL*>(&as_global(dq.w)[j * 256 + w4 * 4]);
  p.b = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.b)[w4 * 4]);
  p.M = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.M)[w4 * 4]);
  p.sh = *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(&as_global(dq.sh)[w4 * 4]);
  // The two GEMMs' per-channel parameters are NOT prefetched here: in the operand-swapped layout they are 16-byte quads
  // (24 registers for both epilogues); each GEMM phase requests its own together with its first weight fragments --
  // the bias is the accumulators' initial value and arrives in the same L2 round trip as the weights.
  (void)lane; (void)wave; (void)pw; (void)cv;
  return p;
}

// la / lm: LDS tables of the block's two LeakyReLUs; addlut: LDS table pair of its ADD.
// MODE: arithmetic flavour (0 exact / 1 gemmlowp double rounding / 2 xnnpack / 3 builtin_mixed), compile-time.
template <int S, int MODE>
__device__ __forceinline__ void resblock_q256(int8_t* QX, int8_t* QD, int8_t* QP, const TileCtx& cx, int d, int off,
                                              const int8_t* la, const int8_t* lm, const DwQ& dq, const ConvQ& pw,
                                              const ConvQ& cv, const AddQ& add, const int32_t* addlut,
                                              const RbqPre& pre, int tb) {
  static_assert(S == 8, "see resblock_q_prefetch");
  constexpr int QS = 288;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int w4 = tid & 63, s = tid >> 6;
  const int R2 = 2 * d;
  LYRA_TSTAMP(tb + 0);
  {  // a = LeakyReLU(X); depthwise over [a(t-2d), a(t-d), a(t)]; the ring row t-2d is replaced by a(t)
    const int base = (cx.sphase[s] * 2) % R2;
    const uint32_t hp = cx.soff(s) + (uint32_t)(off + w4 * 4);
    const bool valid = cx.valid(s);
    // channel e's three taps side by side in one dword -- (tap0, tap1, tap2, 0): two v_perm_b32 -- so that the 3-tap sum is ONE
    // v_dot4_i32_i8 per channel and row (was three sign extensions + three multiply-adds; integer arithmetic, identical)
    int wq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) wq[e] = taps3(pre.ww[0], pre.ww[1], pre.ww[2], e);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int a = lut8w(la, *reinterpret_cast<const int*>(&QX[(t * S + s) * QS + w4 * 4]));
      if (valid) *cx.at<int>(hp + (uint32_t)((base + t) * 256)) = a;
      int o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int acc = __builtin_amdgcn_sdot4(taps3(pre.h[t][0], pre.h[t][1], a, e), wq[e], pre.b[e], false);
        o[e] = conv_code<conv_flavour<MODE, false>()>(acc, pre.M[e], pre.sh[e], dq.zout);             // depthwise
      }
      *reinterpret_cast<int*>(&QD[(t * S + s) * QS + w4 * 4]) = pack8(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  LYRA_TSTAMP(tb + 1);
  const int row = lane & 15;               // activation row (t, s) this lane's C column belongs to
  const int ch0 = wave * 32 + q * 4;       // its four output channels of N tile j: ch0 + 16 * j + e
  {  // pointwise 256 -> 256, int8 LeakyReLU -> QP
    i32x4 acc[2] = {chan_quad(pw.b, wave * 2), chan_quad(pw.b, wave * 2 + 1)};
    const i32x4 pM[2] = {chan_quad(pw.M, wave * 2), chan_quad(pw.M, wave * 2 + 1)};
    const i32x4 psh[2] = {chan_quad(pw.sh, wave * 2), chan_quad(pw.sh, wave * 2 + 1)};
    auto aoff = [&](int c) { return m * QS + c * 64 + q * 16; };
    gemm_i8_t<2, 4>(QD, aoff, pw.w + (wave * 2) * 4 * 64, acc);
#ifdef LYRA_TIMING
    if (tb == 94) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][3])); LYRA_TSTAMP(122); }   // GEMM done (results awaited)
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int r8[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) r8[e] = lut8(lm, conv_code<conv_flavour<MODE, true>()>(acc[j][e], pM[j][e], psh[j][e], pw.zout));   // pointwise: ungrouped
      *reinterpret_cast<int*>(&QP[row * QS + ch0 + 16 * j]) = pack8(r8[0], r8[1], r8[2], r8[3]);
    }
  }
  __syncthreads();
  LYRA_TSTAMP(tb + 2);
  {  // grouped 1x1 (4 groups of 64 -> 64), int8 ADD with the residual stream
    i32x4 acc[2] = {chan_quad(cv.b, wave * 2), chan_quad(cv.b, wave * 2 + 1)};
    const i32x4 cM[2] = {chan_quad(cv.M, wave * 2), chan_quad(cv.M, wave * 2 + 1)};
    const i32x4 csh[2] = {chan_quad(cv.sh, wave * 2), chan_quad(cv.sh, wave * 2 + 1)};
    const int g = wave >> 1;
    auto aoff = [&](int c) { return m * QS + g * 64 + q * 16; };
    gemm_i8_t<2, 1>(QP, aoff, cv.w + (wave * 2) * 64, acc);
#ifdef LYRA_TIMING
    if (tb == 94) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][3])); LYRA_TSTAMP(123); }
#endif
    int xw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) xw[j] = *reinterpret_cast<const int*>(&QX[row * QS + ch0 + 16 * j]);   // residual, 4 channels
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c8 = conv_code<conv_flavour<MODE, false>()>(acc[j][e], cM[j][e], csh[j][e], cv.zout);   // 1x1, 4 groups
        if constexpr (MODE == 2) o[e] = xnn_add(c8, sx8(xw[j], e), add);   // two multiply-adds and a shift: no table
        else o[e] = add_q_lut(addlut, c8, sx8(xw[j], e), add);
      }
      *reinterpret_cast<int*>(&QX[row * QS + ch0 + 16 * j]) = pack8(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  LYRA_TSTAMP(tb + 3);
}

}  // namespace lyra
