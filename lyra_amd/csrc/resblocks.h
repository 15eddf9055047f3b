// resblocks.h -- the fp32 residual-block triplets shared by encoder and decoder stages.
//
// One residual block: a = lrelu(x); d = depthwise_k3_dilated(concat(history, a));
// p = lrelu(pointwise(d)); c = conv1x1(p) (grouped at 128 channels); x <- c + x.  Three blocks with
// dilation 1, 3, 9.  X (residual stream) and one scratch matrix live in LDS for the whole triplet;
// rows are (t, s) -> t*S + s.  The GEMM phases run at the MFMA issue rate; everything else (depthwise,
// epilogues, history traffic, barriers) is hidden by keeping three small workgroups resident per CU.
#pragma once
#include "kernels.h"

namespace lyra {

struct TileCtx {
  uint8_t* state;      // the kernel's own state region: [stream][stride] (state_layout.h)
  const int* sids;     // LDS: stream id of each tile slot
  const int* sphase;   // LDS: frame phase of each tile slot (may be null when no ring is used)
  int nvalid;          // slots < nvalid are real streams
  int stride;          // bytes per stream in this region
  // A stream id of -1 masks a slot (DTX: the hop is noise, the encoder must not run for that stream,
  // lyra_encoder.cc:131-141): it reads stream 0's state like a tail slot reads the last stream's, and writes nothing.
#ifdef LYRA_STATE_ALIAS   // timing experiment only (results are wrong): every stream uses one of 64 state slots -- the
                          // per-stream state never leaves L2, so what the HBM state traffic costs shows as the difference
  __device__ __forceinline__ uint8_t* sbase(int s) const { return state + (size_t)(max(sids[s], 0) & 63) * stride; }
#else
  __device__ __forceinline__ uint8_t* sbase(int s) const { return state + (size_t)max(sids[s], 0) * stride; }
#endif
  __device__ __forceinline__ bool valid(int s) const { return s < nvalid && sids[s] >= 0; }
  // The same address as sbase(s) + byte_off, formed as UNIFORM base + 32-bit per-lane offset (a region is at most
  // max_streams x 15 KB < 4 GB): the access takes the scalar-base form `global_load v, v_off, s[base:base+1]` and the
  // per-lane address arithmetic is 32-bit (one v_add_u32 where the 64-bit form needs v_lshl_add_u64 / add + addc pairs).
  __device__ __forceinline__ uint32_t soff(int s) const {
#ifdef LYRA_STATE_ALIAS
    return (uint32_t)(max(sids[s], 0) & 63) * (uint32_t)stride;
#else
    return (uint32_t)max(sids[s], 0) * (uint32_t)stride;
#endif
  }
  template <class T>
  __device__ __forceinline__ T LYRA_GLOBAL* at(uint32_t byte_off) const {
    return (T LYRA_GLOBAL*)((uint8_t LYRA_GLOBAL*)state + byte_off);
  }
};

// Requests one word of every 128-byte line of [off, off + bytes) of the tile's S streams (one load per thread): state a
// LATER phase of the kernel reads with a dependent load in the middle of its chain (a strided conv's carried rows, a
// transposed conv's overlap tail) is then on its way to the XCD's L2 while the first phases run.  The returned token
// goes to state_touch_sink at the end of the kernel so that the loads stay in the program without being waited for.
template <int S, int NT>
__device__ __forceinline__ uint32_t state_touch(const TileCtx& cx, int off, int bytes) {
  const int l0 = off >> 7, n = ((off + bytes - 1) >> 7) - l0 + 1;
  const int idx = threadIdx.x;
  uint32_t tok = 0;
  if (idx < S * n) {
    const int s = idx / n, l = idx - s * n;
    tok = *cx.at<const uint32_t>(cx.soff(s) + (uint32_t)((l0 + l) << 7));
  }
  return tok;
}
__device__ __forceinline__ void state_touch_sink(uint32_t tok, uint8_t* state, int B) {
  if (tok == 0x9E3779B9u && B == -0x5EED) state[0] = (uint8_t)tok;   // never true
}

// ---- 64 channels x 20 rows; S streams per workgroup of NT threads: (S, NT) = (4, 256) or (8, 512) -------------
// T = 20 >= 2*dilation, so histories are simply replaced.  GEMM [20*S rows] x 64 x 64: wave = (N tile wn = wave & 3,
// M group wm = wave >> 2), 5 M tiles per wave.  The residual stream X lives in REGISTERS (MFMA C layout):
// xr[i][0][e] = X[row (wm*5+i)*16 + 4q + e][channel wn*16 + (lane&15)].  Only ONE LDS matrix A[20][S][72] is needed
// (it carries lrelu(X), then the depthwise output, then the pointwise output in turn), so a tile of S = 4 streams
// takes < 30 KB and four workgroups fit on a CU -- the whole B = 4096 batch is resident in one wave of workgroups.
// The depthwise conv runs on (row, channel quad) items with 16-byte LDS / history accesses; the residual add is a
// register add.
// LYRA_SWAP64 (round 6): the GEMMs run operand-swapped (lyra_dev.h gemm_f32_core SWAP) -- xr[i][0][e] =
// X[row (wm*5+i)*16 + (lane&15)][physical channel wn*16 + 4q + e], and every epilogue is one 16-byte LDS store per C tile.
template <int S, int NT>
__device__ __forceinline__ void resblocks64r(f32x4 (&xr)[5][1], float* A, const TileCtx& cx, const DwF* dws,
                                             const ConvF* pws, const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int CS = 72;
  constexpr bool SW = LYRA_SWAP64 != 0;
  static_assert(20 * S / 16 == 5 * (NT / 256), "5 M tiles per wave");
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    // The per-thread index math is recomputed inside the loop on purpose: hoisted out of it (LICM) its ~25
    // loop-invariant address registers do not fit next to the resident X under 128 VGPRs and get spilled.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int wn = wave & 3, wm = wave >> 2;
    const int ncol = wn * 16 + (lane & 15);
    const int pcol = at16(ncol);
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    LYRA_TSTAMP(10 + r * 8 + 0);
    // 0. request the depthwise parameters and the history rows this thread will need (L2 / HBM latency overlaps
    //    the a-write and the barrier).  The depthwise conv runs on (row, channel quad) items -- 5 per thread, rows
    //    rq + k * RSTEP -- so history and LDS traffic are 16-byte accesses (a few dwordx4 loads per thread instead
    //    of up to ~100 dword loads in the MFMA C layout, which saturated the CU's address path).
    constexpr int RSTEP = NT / 16;
    const int p4 = tid & 15, rq = tid >> 4;
    const float LYRA_GLOBAL* dww = as_global(dws[r].w);
    // RSTEP is a multiple of S: a thread's five items are rows t = tq + k * TSTEP of ONE stream sq.  Tap 0 of row t
    // reads history row 2d + (t - 2d) = t (when t < 2d), tap 1 reads history row t + d (when t < d): two base
    // pointers + immediate offsets.
    static_assert(RSTEP % S == 0, "one stream per thread");
    constexpr int TSTEP = RSTEP / S;
    const int sq = rq & (S - 1), tq = rq / S;
    const uint32_t hb0 = cx.soff(sq) + (uint32_t)(off + (tq * 64 + p4 * 4) * 4);
    const uint32_t hb1 = hb0 + (uint32_t)(d * 256);
    f32x4 h0[5], h1[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int t = tq + k * TSTEP;
      h0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      h1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t < 2 * d) h0[k] = *cx.at<const f32x4>(hb0 + k * TSTEP * 256);
      if (t < d) h1[k] = *cx.at<const f32x4>(hb1 + k * TSTEP * 256);
    }
    // 1. a = lrelu(X) -> A
#pragma unroll
    for (int i = 0; i < 5; ++i)
{
      const f32x4 a4 = lrelu4(xr[i][0]);
      if constexpr (SW) {
        *reinterpret_cast<f32x4*>(&A[((wm * 5 + i) * 16 + m) * CS + wn * 16 + q * 4]) = a4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) A[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = a4[e];
      }
    }
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 1);
    // 2. depthwise k3 (dilation d), one tap at a time over the thread's 5 items: the chain value replaces the
    //    tap-0 history register, so the live set stays small (xr + two history sets) under the 128-VGPR budget
    f32x4 dreg[5];
    {
      const f32x4 w0 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + p4 * 4);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int t0 = tq + k * TSTEP - 2 * d;
        const f32x4 v0 = t0 >= 0 ? *reinterpret_cast<const f32x4*>(&A[(t0 * S + sq) * CS + p4 * 4]) : h0[k];
        // the chain starts from the bias (XNNPACK's DWCONV order), and every depthwise bias of these graphs is +0.0
        // (bias-free layers; model.hip refuses a container where that is not so)
        dreg[k] = fma4(v0, w0, (f32x4){0.f, 0.f, 0.f, 0.f});
      }
      const f32x4 w1 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 64 + p4 * 4);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int t1 = tq + k * TSTEP - d;
        const f32x4 v1 = t1 >= 0 ? *reinterpret_cast<const f32x4*>(&A[(t1 * S + sq) * CS + p4 * 4]) : h1[k];
        dreg[k] = fma4(v1, w1, dreg[k]);
      }
      const f32x4 w2 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 128 + p4 * 4);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(&A[(rq + k * RSTEP) * CS + p4 * 4]);
        dreg[k] = fma4(v2, w2, dreg[k]);
      }
    }
    LYRA_TSTAMP(10 + r * 8 + 2);
    __syncthreads();  // every thread has consumed its history rows and read A
    // 3. new history = last R2 rows of a (T = 20 >= R2), and the depthwise output -> A: an item (row, channel quad) is
    //    read and rewritten by its owner only, so the two need no barrier between them
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      f32x4* item = reinterpret_cast<f32x4*>(&A[(rq + k * RSTEP) * CS + p4 * 4]);
      const int j = tq + k * TSTEP - (20 - R2);
      if (j >= 0 && cx.valid(sq)) *cx.at<f32x4>(cx.soff(sq) + (uint32_t)(off + (j * 64 + p4 * 4) * 4)) = *item;
      *item = dreg[k];
    }
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 4);
    auto aoff = [&](int i, int c) { return ((wm * 5 + i) * 16 + m) * CS + c * 16 + q * 4; };
    WPre<1, 1> cv_pre;
    {  // 4. pointwise 64 -> 64, LeakyReLU -> A
      f32x4 acc[5][1];
      gemm_f32_bias<5, 1, 4, 4, gemm_pf<5, 1>(), SW>(A, aoff, pws[r].w + wn * 4 * 64, pws[r].b, wn * 16, acc);
      LYRA_TSTAMP(10 + r * 8 + 5);
      // the 1x1 conv's bias + first weight chunk, requested ahead of the barrier in front of it (see resblocks128; round 5:
      // +1.0 % on the whole step, +2.1 % at 1,024 streams).  The pointwise GEMM's own request stays behind its barrier: held
      // across the depthwise phase it costs spills at the 128-VGPR cap.
      cv_pre = gemm_f32_wprefetch<1, 4, 4, 1, SW>(cvs[r].w + wn * 4 * 64, cvs[r].b, wn * 16);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 5; ++i)
{
        const f32x4 a4 = lrelu4(acc[i][0]);
        if constexpr (SW) {
          *reinterpret_cast<f32x4*>(&A[((wm * 5 + i) * 16 + m) * CS + wn * 16 + q * 4]) = a4;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) A[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = a4[e];
        }
      }
      __syncthreads();
      LYRA_TSTAMP(10 + r * 8 + 6);
    }
    {  // 5. 1x1 conv 64 -> 64 + residual (registers)
      f32x4 acc[5][1];
      gemm_f32_pre<5, 1, 4, 4, gemm_pf<5, 1>(), SW>(A, aoff, cvs[r].w + wn * 4 * 64, cv_pre, acc);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[i][0][e] = acc[i][0][e] + xr[i][0][e];
    }
    LYRA_TSTAMP(10 + r * 8 + 7);
    __syncthreads();  // all waves are done reading A before the next block overwrites it
  }
}

// ---- 128 channels x 4 rows; S = 8 streams, NT = 256 or 512 threads ----------------------------------------------
// X: [4][S][136] floats, D: same.  Dilation 1 keeps the last two rows; dilations 3 and 9 use ring histories of
// 6 / 18 rows (T = 4 new rows per step at slot (phase*4 + t) mod R).
// GEMM [4*S rows] x 128 x 128: 2 M tiles per wave, 8 N tiles spread over the NT/64 waves.
// Thread (s, p4, half) owns channel quad p4 of stream s for RPT = 1024/NT rows (all four with 256 threads, rows
// {0,1} / {2,3} with 512): LeakyReLU, the depthwise conv and the history update are thread-local, every history tap
// is a 16-byte global load, and the loads of block r+1 are issued behind block r's LDS-only phases.
template <int RPT>
struct Hist128 { f32x4 h[RPT][2]; };   // [own row][tap 0 (t-2d), tap 1 (t-d)], valid where the tap predates the frame

template <int S, int NT>
__device__ __forceinline__ Hist128<1024 / NT> hist128_prefetch(const TileCtx& cx, int d, int off) {
  constexpr int RPT = 1024 / NT;
  const int tid = threadIdx.x, p4 = tid & 31, s = (tid >> 5) & (S - 1), row0 = (tid >> 8) * RPT;
  const int R2 = 2 * d;
  const bool ring = R2 > 4;
  const int base = ring ? (cx.sphase[s] * 4) % R2 : 0;
  const uint32_t hp = cx.soff(s) + (uint32_t)(off + p4 * 16);
  Hist128<RPT> H;
#pragma unroll
  for (int k = 0; k < RPT; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tau = row0 + k - (2 - j) * d;
      H.h[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (tau < 0) {
        int row = R2 + tau;
        if (ring) { row = base + tau + R2; row = row >= R2 ? row - R2 : row; }
        H.h[k][j] = *cx.at<const f32x4>(hp + (uint32_t)(row * 512));
      }
    }
  return H;
}

// H: hist128_prefetch(cx, 1, off0), requested by the caller together with the stage input.
// P: a third [4][S][136] matrix for the pointwise output -- written while other waves may still read D in their GEMM,
// so no barrier is needed between the two.
template <int S, int NT>
__device__ __forceinline__ void resblocks128(float* X, float* D, float* P, const TileCtx& cx, const DwF* dws,
                                             const ConvF* pws, const ConvF* cvs, int off0, int off1, int off2,
                                             Hist128<1024 / NT> H) {
  static_assert(S == 8 && (NT == 256 || NT == 512), "thread <-> (stream, channel quad, row half) mapping");
  constexpr int CS = 136, NTW = 8 / (NT / 64), MT = 2, RPT = 1024 / NT;
  constexpr bool SW = LYRA_SWAP128 != 0;   // operand-swapped GEMMs (lyra_dev.h): 16-byte epilogue accesses
#ifndef LYRA_PF128_EXTRA
#define LYRA_PF128_EXTRA 0   // experiment: weight / operand prefetch distance of the 128-channel blocks' GEMMs, in K chunks beyond the default
#endif
  constexpr int PF = gemm_pf<MT, NTW>() + LYRA_PF128_EXTRA;
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // keep the index math inside the loop (see resblocks64r)
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int p4 = tid & 31, s = (tid >> 5) & (S - 1), row0 = (tid >> 8) * RPT;
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    const bool ring = R2 > 4;
    LYRA_TSTAMP(40 + r * 8 + 0);
    {  // a = lrelu(X); depthwise [a(t-2d), a(t-d), a(t)]; history update -- all on this thread's own quad.
       // A history slot read here (prefetched) may be rewritten by the thread owning the other row half; its
       // prefetch retired before the previous block's second GEMM could fetch weights (vmcnt is in order), i.e.
       // before the barrier that precedes this phase.
      const float LYRA_GLOBAL* dww = as_global(dws[r].w) + p4 * 4;
      const f32x4 w0 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww);
      const f32x4 w1 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 128);
      const f32x4 w2 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 256);
      const uint32_t hp = cx.soff(s) + (uint32_t)(off + p4 * 16);
      const int base = ring ? (cx.sphase[s] * 4) % R2 : 0;
      const bool valid = cx.valid(s);
      const float* xq = &X[s * CS + p4 * 4];   // this thread's quad of row t at xq[t * S * CS]
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int t = row0 + k, t0 = t - 2 * d, t1 = t - d;
        const f32x4 a = lrelu4(*reinterpret_cast<const f32x4*>(xq + t * S * CS));
        // taps that fall inside the frame are earlier rows of lrelu(X) (only possible for d <= 3)
        f32x4 v0 = H.h[k][0], v1 = H.h[k][1];
        if (t0 >= 0) v0 = lrelu4(*reinterpret_cast<const f32x4*>(xq + t0 * S * CS));
        if (t1 >= 0) v1 = lrelu4(*reinterpret_cast<const f32x4*>(xq + t1 * S * CS));
        f32x4 acc = fma4(v0, w0, (f32x4){0.f, 0.f, 0.f, 0.f});   // the chain starts from the bias, which is +0.0 (see resblocks64r)
        acc = fma4(v1, w1, acc);
        acc = fma4(a, w2, acc);
        *reinterpret_cast<f32x4*>(&D[(t * S + s) * CS + p4 * 4]) = acc;
        if (valid) {
          if (ring) {
            int row = base + t;
            row = row >= R2 ? row - R2 : row;
            *cx.at<f32x4>(hp + (uint32_t)(row * 512)) = a;
          } else if (t >= 2) {
            *cx.at<f32x4>(hp + (uint32_t)((t - 2) * 512)) = a;
          }
        }
      }
    }
    // The first weight chunks + bias of each GEMM are requested ahead of the barrier / elementwise phase in front of it
    // (gemm_f32_wprefetch): the L2 round trip no longer stands between the barrier and the first MFMA.  Round 5: +1.2 % on
    // the whole step at 4,096 streams (enc_s1 -1.1 us, dec_s1 -2.2 us); the same in the 64-channel blocks costs spills.
    const auto pw_pre = gemm_f32_wprefetch<NTW, 8, 8, PF, SW>(pws[r].w + (wave * NTW) * 8 * 64, pws[r].b, wave * NTW * 16);
    WPre<NTW, PF> cv_pre;
    __syncthreads();
    LYRA_TSTAMP(40 + r * 8 + 2);
    {  // pointwise 128 -> 128, LeakyReLU
      f32x4 acc[MT][NTW];
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + c * 16 + q * 4; };
      gemm_f32_pre<MT, NTW, 8, 8, PF, SW>(D, aoff, pws[r].w + (wave * NTW) * 8 * 64, pw_pre, acc);
      cv_pre = gemm_f32_wprefetch<NTW, 4, 4, PF, SW>(cvs[r].w + (wave * NTW) * 4 * 64, cvs[r].b, wave * NTW * 16);
      LYRA_TSTAMP(40 + r * 8 + 3);
      // The next block's history rows.  vmcnt retires in order, so these loads would stall the first weight
      // fetch of a GEMM issued right after them; here they have the two barriers and the LDS-only P write
      // (no younger global load) to land in.
      if (r < 2) H = hist128_prefetch<S, NT>(cx, r == 0 ? 3 : 9, r == 0 ? off1 : off2);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int ncol = (wave * NTW + j) * 16 + (lane & 15);
        const int pcol = at16(ncol);
#pragma unroll
        for (int i = 0; i < MT; ++i)
{
          const f32x4 a4 = lrelu4(acc[i][j]);
          if constexpr (SW) {
            *reinterpret_cast<f32x4*>(&P[(i * 16 + m) * CS + (wave * NTW + j) * 16 + q * 4]) = a4;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) P[(i * 16 + q * 4 + e) * CS + pcol] = a4[e];
          }
        }
      }
      __syncthreads();
      LYRA_TSTAMP(40 + r * 8 + 4);
    }
    {  // grouped 1x1 (2 groups of 64 -> 64) + residual; a wave's N tiles lie in one group
      f32x4 acc[MT][NTW];
      const int g = (wave * NTW) >> 2;
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + g * 64 + c * 16 + q * 4; };
      gemm_f32_pre<MT, NTW, 4, 4, PF, SW>(P, aoff, cvs[r].w + (wave * NTW) * 4 * 64, cv_pre, acc);
      LYRA_TSTAMP(40 + r * 8 + 5);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int ncol = (wave * NTW + j) * 16 + (lane & 15);
        const int pcol = at16(ncol);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if constexpr (SW) {
            f32x4* x = reinterpret_cast<f32x4*>(&X[(i * 16 + m) * CS + (wave * NTW + j) * 16 + q * 4]);
            const f32x4 old = *x;
            f32x4 sum;
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] = acc[i][j][e] + old[e];
            *x = sum;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float* x = &X[(i * 16 + q * 4 + e) * CS + pcol];
              *x = acc[i][j][e] + *x;
            }
          }
        }
      }
    }
    __syncthreads();
    LYRA_TSTAMP(40 + r * 8 + 6);
  }
}

}  // namespace lyra
