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
  uint8_t* state;
  const int* sids;     // LDS: stream id of each tile slot
  const int* sphase;   // LDS: frame phase of each tile slot (may be null when no ring is used)
  int nvalid;          // slots < nvalid are real streams
  __device__ __forceinline__ uint8_t* sbase(int s) const { return state + (size_t)sids[s] * st::BYTES; }
  __device__ __forceinline__ bool valid(int s) const { return s < nvalid; }
};

// ---- 64 channels x 20 rows; S streams per workgroup of NT threads: (S, NT) = (4, 256) or (8, 512) -------------
// X: [20][S][72] floats, D: same shape.  T = 20 >= 2*dilation, so histories are simply replaced.
// GEMM [20*S rows] x 64 x 64: wave = (N tile wn = wave & 3, M group wm = wave >> 2), 5 M tiles per wave.
template <int S, int NT>
__device__ __forceinline__ void resblocks64(float* X, float* D, const TileCtx& cx, const DwF* dws, const ConvF* pws,
                                            const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int CS = 72;
  static_assert(20 * S / 16 == 5 * (NT / 256), "5 M tiles per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;
  const int ncol = wn * 16 + (lane & 15);
  const int pcol = at16(ncol);
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    const DwF dw = dws[r];
    LYRA_TSTAMP(10 + r * 8 + 0);
#ifndef LYRA_ABL_NODW
    for (int idx = tid; idx < 20 * S * 16; idx += NT) {
      int p4 = idx & 15, s = (idx >> 4) & (S - 1), t = (idx >> 4) / S;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int tau = t - (2 - j) * d;
        f32x4 v;
        if (tau >= 0) v = lrelu4(*reinterpret_cast<const f32x4*>(&X[(tau * S + s) * CS + p4 * 4]));
        else v = *reinterpret_cast<const f32x4*>(cx.sbase(s) + off + ((R2 + tau) * 64 + p4 * 4) * 4);
        acc = fma4(v, *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(dw.w) + j * 64 + p4 * 4), acc);
      }
      f32x4 bb = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(dw.b) + p4 * 4);
      *reinterpret_cast<f32x4*>(&D[(t * S + s) * CS + p4 * 4]) = acc + bb;
    }
#endif
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 1);
    for (int idx = tid; idx < R2 * S * 16; idx += NT) {
      int p4 = idx & 15, s = (idx >> 4) & (S - 1), j = (idx >> 4) / S;
      if (cx.valid(s))
        *reinterpret_cast<f32x4*>(cx.sbase(s) + off + (j * 64 + p4 * 4) * 4) =
            lrelu4(*reinterpret_cast<const f32x4*>(&X[((20 - R2 + j) * S + s) * CS + p4 * 4]));
    }
    LYRA_TSTAMP(10 + r * 8 + 2);
    auto aoff = [&](int i, int c) { return ((wm * 5 + i) * 16 + m) * CS + c * 16 + q * 4; };
    {
      f32x4 acc[5][1];
      gemm_f32<5, 1, 4>(D, aoff, pws[r].w + wn * 4 * 64, acc);
      float bias = as_global(pws[r].b)[ncol];
      LYRA_TSTAMP(10 + r * 8 + 3);
      __syncthreads();
      LYRA_TSTAMP(10 + r * 8 + 4);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) D[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = lrelu(acc[i][0][e] + bias);
      __syncthreads();
      LYRA_TSTAMP(10 + r * 8 + 5);
    }
    {
      f32x4 acc[5][1];
      gemm_f32<5, 1, 4>(D, aoff, cvs[r].w + wn * 4 * 64, acc);
      float bias = as_global(cvs[r].b)[ncol];
      LYRA_TSTAMP(10 + r * 8 + 6);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* x = &X[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol];
          *x = (acc[i][0][e] + bias) + *x;
        }
    }
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 7);
  }
}

// ---- same triplet with the residual stream X resident in REGISTERS (MFMA C layout) ------------------------------
// xr[i][0][e] = X[row (wm*5+i)*16 + 4q + e][channel wn*16 + (lane&15)].  Only ONE LDS matrix A[20][S][72] is needed
// (it carries lrelu(X), then the depthwise output, then the pointwise output in turn), so a tile of S = 4 streams
// takes < 30 KB and four workgroups fit on a CU -- the whole B = 4096 batch is resident in one wave of workgroups.
// The depthwise conv runs in the C layout: each lane produces the (row, channel) elements it owns from three
// LDS rows; the residual add is a register add.
template <int S, int NT>
__device__ __forceinline__ void resblocks64r(f32x4 (&xr)[5][1], float* A, const TileCtx& cx, const DwF* dws,
                                             const ConvF* pws, const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int CS = 72;
  static_assert(20 * S / 16 == 5 * (NT / 256), "5 M tiles per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;
  const int ncol = wn * 16 + (lane & 15);
  const int pcol = at16(ncol);
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    LYRA_TSTAMP(10 + r * 8 + 0);
    // 0. request the per-channel depthwise parameters and the history rows this lane will need
    //    (L2 / HBM latency overlaps the a-write and the barrier)
    const float LYRA_GLOBAL* dww = as_global(dws[r].w);
    const float w0 = dww[pcol], w1 = dww[64 + pcol], w2 = dww[128 + pcol];
    const float bb = as_global(dws[r].b)[pcol];
    f32x4 h0[5], h1[5];
    const float LYRA_GLOBAL* hist[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = (S >= 4) ? (e & (S - 1)) : 0;  // with S >= 4 a lane's four rows are four streams at one t
      hist[e] = as_global(reinterpret_cast<const float*>(cx.sbase(s) + off));
    }
    static_assert(S == 4 || S == 8, "row -> (t, s) mapping below");
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int R = (wm * 5 + i) * 16 + q * 4 + e, t = R / S, s = R & (S - 1);
        const float LYRA_GLOBAL* hp = (S == 4) ? hist[e] : as_global(reinterpret_cast<const float*>(cx.sbase(s) + off));
        const int t0 = t - 2 * d, t1 = t - d;
        h0[i][e] = t0 < 0 ? hp[(R2 + t0) * 64 + pcol] : 0.f;
        h1[i][e] = t1 < 0 ? hp[(R2 + t1) * 64 + pcol] : 0.f;
      }
    // 1. a = lrelu(X) -> A
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) A[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = lrelu(xr[i][0][e]);
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 1);
    // 2. depthwise k3 (dilation d) for the elements this lane owns
    f32x4 dreg[5];
    {
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int R = (wm * 5 + i) * 16 + q * 4 + e, t = R / S, s = R & (S - 1);
          const int t0 = t - 2 * d, t1 = t - d;
          float v0 = t0 >= 0 ? A[(t0 * S + s) * CS + pcol] : h0[i][e];
          float v1 = t1 >= 0 ? A[(t1 * S + s) * CS + pcol] : h1[i][e];
          float v2 = A[R * CS + pcol];
          float acc = __builtin_fmaf(v0, w0, 0.f);
          acc = __builtin_fmaf(v1, w1, acc);
          acc = __builtin_fmaf(v2, w2, acc);
          dreg[i][e] = acc + bb;
        }
    }
    LYRA_TSTAMP(10 + r * 8 + 2);
    __syncthreads();  // every lane has read its history rows and A
    for (int idx = tid; idx < R2 * S * 16; idx += NT) {  // new history = last R2 rows of a (T = 20 >= R2)
      int p4 = idx & 15, s = (idx >> 4) & (S - 1), j = (idx >> 4) / S;
      if (cx.valid(s))
        *reinterpret_cast<f32x4*>(cx.sbase(s) + off + (j * 64 + p4 * 4) * 4) =
            *reinterpret_cast<const f32x4*>(&A[((20 - R2 + j) * S + s) * CS + p4 * 4]);
    }
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 3);
    // 3. depthwise out -> A
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) A[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = dreg[i][e];
    __syncthreads();
    LYRA_TSTAMP(10 + r * 8 + 4);
    auto aoff = [&](int i, int c) { return ((wm * 5 + i) * 16 + m) * CS + c * 16 + q * 4; };
    {  // 4. pointwise 64 -> 64, LeakyReLU -> A
      f32x4 acc[5][1];
      float bias = as_global(pws[r].b)[ncol];
      gemm_f32<5, 1, 4>(A, aoff, pws[r].w + wn * 4 * 64, acc);
      LYRA_TSTAMP(10 + r * 8 + 5);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) A[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = lrelu(acc[i][0][e] + bias);
      __syncthreads();
      LYRA_TSTAMP(10 + r * 8 + 6);
    }
    {  // 5. 1x1 conv 64 -> 64 + residual (registers)
      f32x4 acc[5][1];
      float bias = as_global(cvs[r].b)[ncol];
      gemm_f32<5, 1, 4>(A, aoff, cvs[r].w + wn * 4 * 64, acc);
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[i][0][e] = (acc[i][0][e] + bias) + xr[i][0][e];
    }
    LYRA_TSTAMP(10 + r * 8 + 7);
    __syncthreads();  // all waves are done reading A before the next block overwrites it
  }
}

// ---- 128 channels x 4 rows; (S, NT) = (8, 256) or (16, 512) -------------------------------------------------------
// X: [4][S][136] floats, D: same.  Dilation 1 keeps the last two rows; dilations 3 and 9 use ring histories of
// 6 / 18 rows (T = 4 new rows per step at slot (phase*4 + t) mod R).
// GEMM [4*S rows] x 128 x 128: MT = S/4 M tiles per wave, 8 N tiles spread over the NT/64 waves.
template <int S, int NT>
__device__ __forceinline__ void resblocks128(float* X, float* D, const TileCtx& cx, const DwF* dws,
                                             const ConvF* pws, const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int CS = 136, NW = NT / 64, NTW = 8 / NW, MT = (4 * S) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    const bool ring = R2 > 4;
    const DwF dw = dws[r];
    for (int idx = tid; idx < 4 * S * 32; idx += NT) {
      int p4 = idx & 31, s = (idx >> 5) & (S - 1), t = (idx >> 5) / S;
      int base = ring ? (cx.sphase[s] * 4) % R2 : 0;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int tau = t - (2 - j) * d;
        f32x4 v;
        if (tau >= 0) {
          v = lrelu4(*reinterpret_cast<const f32x4*>(&X[(tau * S + s) * CS + p4 * 4]));
        } else {
          int row = R2 + tau;
          if (ring) { row = base + tau + R2; row = row >= R2 ? row - R2 : row; }
          v = *reinterpret_cast<const f32x4*>(cx.sbase(s) + off + (row * 128 + p4 * 4) * 4);
        }
        acc = fma4(v, *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(dw.w) + j * 128 + p4 * 4), acc);
      }
      f32x4 bb = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(dw.b) + p4 * 4);
      *reinterpret_cast<f32x4*>(&D[(t * S + s) * CS + p4 * 4]) = acc + bb;
    }
    __syncthreads();
    {
      const int nrows = ring ? 4 : 2;
      for (int idx = tid; idx < nrows * S * 32; idx += NT) {
        int p4 = idx & 31, s = (idx >> 5) & (S - 1), j = (idx >> 5) / S;
        int src_t, row;
        if (ring) {
          int base = (cx.sphase[s] * 4) % R2;
          src_t = j; row = base + j; row = row >= R2 ? row - R2 : row;
        } else {
          src_t = 2 + j; row = j;
        }
        if (cx.valid(s))
          *reinterpret_cast<f32x4*>(cx.sbase(s) + off + (row * 128 + p4 * 4) * 4) =
              lrelu4(*reinterpret_cast<const f32x4*>(&X[(src_t * S + s) * CS + p4 * 4]));
      }
    }
    {  // pointwise 128 -> 128, LeakyReLU
      f32x4 acc[MT][NTW];
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + c * 16 + q * 4; };
      float biasv[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) biasv[j] = as_global(pws[r].b)[(wave * NTW + j) * 16 + (lane & 15)];
      gemm_f32<MT, NTW, 8>(D, aoff, pws[r].w + (wave * NTW) * 8 * 64, acc);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int ncol = (wave * NTW + j) * 16 + (lane & 15);
        const float bias = biasv[j];
        const int pcol = at16(ncol);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) D[(i * 16 + q * 4 + e) * CS + pcol] = lrelu(acc[i][j][e] + bias);
      }
      __syncthreads();
    }
    {  // grouped 1x1 (2 groups of 64 -> 64) + residual; a wave's N tiles lie in one group
      f32x4 acc[MT][NTW];
      const int g = (wave * NTW) >> 2;
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + g * 64 + c * 16 + q * 4; };
      float biasv[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) biasv[j] = as_global(cvs[r].b)[(wave * NTW + j) * 16 + (lane & 15)];
      gemm_f32<MT, NTW, 4>(D, aoff, cvs[r].w + (wave * NTW) * 4 * 64, acc);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int ncol = (wave * NTW + j) * 16 + (lane & 15);
        const float bias = biasv[j];
        const int pcol = at16(ncol);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float* x = &X[(i * 16 + q * 4 + e) * CS + pcol];
            *x = (acc[i][j][e] + bias) + *x;
          }
      }
    }
    __syncthreads();
  }
}

}  // namespace lyra
