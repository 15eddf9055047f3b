// resblocks.h -- the fp32 residual-block triplets shared by encoder and decoder stages.
//
// One residual block: a = lrelu(x); d = depthwise_k3_dilated(concat(history, a));
// p = lrelu(pointwise(d)); c = conv1x1(p) (grouped at 128 channels); x <- c + x.  Three blocks with
// dilation 1, 3, 9.  X (residual stream) and one scratch matrix live in LDS for the whole triplet.
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

// ---- 64 channels x 20 rows, 8 streams per workgroup, 512 threads ----------------------------------
// X: [20][8][72] floats (row = t*8 + s), D: same shape.  T = 20 >= 2*dilation, histories are replaced.
__device__ __forceinline__ void resblocks64(float* X, float* D, const TileCtx& cx, const DwF* dws, const ConvF* pws,
                                            const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int S = 8, CS = 72, NT = 512;
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
    for (int idx = tid; idx < 20 * S * 16; idx += NT) {
      int p4 = idx & 15, s = (idx >> 4) & 7, t = idx >> 7;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int tau = t - (2 - j) * d;
        f32x4 v;
        if (tau >= 0) v = lrelu4(*reinterpret_cast<const f32x4*>(&X[(tau * S + s) * CS + p4 * 4]));
        else v = *reinterpret_cast<const f32x4*>(cx.sbase(s) + off + ((R2 + tau) * 64 + p4 * 4) * 4);
        acc = fma4(v, *reinterpret_cast<const f32x4*>(&dw.w[j * 64 + p4 * 4]), acc);
      }
      f32x4 bb = *reinterpret_cast<const f32x4*>(&dw.b[p4 * 4]);
      *reinterpret_cast<f32x4*>(&D[(t * S + s) * CS + p4 * 4]) = acc + bb;
    }
    __syncthreads();
    for (int idx = tid; idx < R2 * S * 16; idx += NT) {
      int p4 = idx & 15, s = (idx >> 4) & 7, j = idx >> 7;
      if (cx.valid(s))
        *reinterpret_cast<f32x4*>(cx.sbase(s) + off + (j * 64 + p4 * 4) * 4) =
            lrelu4(*reinterpret_cast<const f32x4*>(&X[((20 - R2 + j) * S + s) * CS + p4 * 4]));
    }
    auto aoff = [&](int i, int c) { return ((wm * 5 + i) * 16 + m) * CS + c * 16 + q * 4; };
    {
      f32x4 acc[5][1];
      gemm_f32<5, 1, 4>(D, aoff, pws[r].w + wn * 4 * 64, acc);
      float bias = pws[r].b[ncol];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) D[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol] = lrelu(acc[i][0][e] + bias);
      __syncthreads();
    }
    {
      f32x4 acc[5][1];
      gemm_f32<5, 1, 4>(D, aoff, cvs[r].w + wn * 4 * 64, acc);
      float bias = cvs[r].b[ncol];
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* x = &X[((wm * 5 + i) * 16 + q * 4 + e) * CS + pcol];
          *x = (acc[i][0][e] + bias) + *x;
        }
    }
    __syncthreads();
  }
}

// ---- 128 channels x 4 rows, 16 streams per workgroup, 512 threads -----------------------------------
// X: [4][16][136] floats (row = t*16 + s), D: same.  Dilation 1 keeps the last two rows; dilations 3 and
// 9 use ring histories of 6 / 18 rows (T = 4 new rows per step at slot (phase*4 + t) mod R).
__device__ __forceinline__ void resblocks128(float* X, float* D, const TileCtx& cx, const DwF* dws,
                                             const ConvF* pws, const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int S = 16, CS = 136, NT = 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int ncol = wave * 16 + (lane & 15);
  const int pcol = at16(ncol);
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    const bool ring = R2 > 4;
    const DwF dw = dws[r];
    for (int idx = tid; idx < 4 * S * 32; idx += NT) {
      int p4 = idx & 31, s = (idx >> 5) & 15, t = idx >> 9;
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
        acc = fma4(v, *reinterpret_cast<const f32x4*>(&dw.w[j * 128 + p4 * 4]), acc);
      }
      f32x4 bb = *reinterpret_cast<const f32x4*>(&dw.b[p4 * 4]);
      *reinterpret_cast<f32x4*>(&D[(t * S + s) * CS + p4 * 4]) = acc + bb;
    }
    __syncthreads();
    {
      const int nrows = ring ? 4 : 2;
      for (int idx = tid; idx < nrows * S * 32; idx += NT) {
        int p4 = idx & 31, s = (idx >> 5) & 15, j = idx >> 9;
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
    {
      f32x4 acc[4][1];
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + c * 16 + q * 4; };
      gemm_f32<4, 1, 8>(D, aoff, pws[r].w + wave * 8 * 64, acc);
      float bias = pws[r].b[ncol];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) D[(i * 16 + q * 4 + e) * CS + pcol] = lrelu(acc[i][0][e] + bias);
      __syncthreads();
    }
    {
      f32x4 acc[4][1];
      const int g = wave >> 2;
      auto aoff = [&](int i, int c) { return (i * 16 + m) * CS + g * 64 + c * 16 + q * 4; };
      gemm_f32<4, 1, 4>(D, aoff, cvs[r].w + wave * 4 * 64, acc);
      float bias = cvs[r].b[ncol];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float* x = &X[(i * 16 + q * 4 + e) * CS + pcol];
          *x = (acc[i][0][e] + bias) + *x;
        }
    }
    __syncthreads();
  }
}

// ---- int8 residual block at 256 channels x 2 rows, 16 streams per workgroup, 512 threads -----------
// QX residual stream, QA/QD/QP scratch, all [2][S][288] int8.  Ring history of R2 = 2*d rows, T = 2.
__device__ __forceinline__ int sx8(int w, int i) { return (int)(int8_t)(w >> (8 * i)); }
__device__ __forceinline__ int pack8(int a, int b, int c, int d) {
  return (a & 255) | ((b & 255) << 8) | ((c & 255) << 16) | ((d & 255) << 24);
}

template <int S>
__device__ __forceinline__ void resblock_q256(int8_t* QX, int8_t* QA, int8_t* QD, int8_t* QP, const TileCtx& cx,
                                              int d, int off, const LreluQ& la, const LreluQ& lm, const DwQ& dq,
                                              const ConvQ& pw, const ConvQ& cv, const AddQ& add, int mode) {
  // rows = (t, s) -> t * S + s, T = 2; S = 16: two M tiles, S = 8: one.
  constexpr int QS = 288, NT = 512, MT = (2 * S) / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int R2 = 2 * d;
  for (int idx = tid; idx < 2 * S * 64; idx += NT) {
    int w4 = idx & 63, rs = idx >> 6;
    int w = *reinterpret_cast<const int*>(&QX[rs * QS + w4 * 4]);
    *reinterpret_cast<int*>(&QA[rs * QS + w4 * 4]) =
        pack8(lrelu_q(sx8(w, 0), la), lrelu_q(sx8(w, 1), la), lrelu_q(sx8(w, 2), la), lrelu_q(sx8(w, 3), la));
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S * 64; idx += NT) {
    int w4 = idx & 63, s = (idx >> 6) & (S - 1), t = (idx >> 6) / S;
    int base = (cx.sphase[s] * 2) % R2;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int tau = t - (2 - j) * d;
      int w;
      if (tau >= 0) {
        w = *reinterpret_cast<const int*>(&QA[(tau * S + s) * QS + w4 * 4]);
      } else {
        int row = base + tau + R2;
        row = row >= R2 ? row - R2 : row;
        w = *reinterpret_cast<const int*>(cx.sbase(s) + off + row * 256 + w4 * 4);
      }
      int ww = *reinterpret_cast<const int*>(&dq.w[j * 256 + w4 * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += (sx8(w, e) - dq.zin) * sx8(ww, e);
    }
    int o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int c = w4 * 4 + e;
      o[e] = clamp8(requant(acc[e] + dq.b[c], dq.M[c], dq.sh[c], mode) + dq.zout);
    }
    *reinterpret_cast<int*>(&QD[(t * S + s) * QS + w4 * 4]) = pack8(o[0], o[1], o[2], o[3]);
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S * 64; idx += NT) {
    int w4 = idx & 63, s = (idx >> 6) & (S - 1), t = (idx >> 6) / S;
    int row = (cx.sphase[s] * 2) % R2 + t;
    row = row >= R2 ? row - R2 : row;
    if (cx.valid(s))
      *reinterpret_cast<int*>(cx.sbase(s) + off + row * 256 + w4 * 4) =
          *reinterpret_cast<const int*>(&QA[(t * S + s) * QS + w4 * 4]);
  }
  {
    i32x4 acc[MT][2];
    auto aoff = [&](int i, int c) { return (i * 16 + m) * QS + c * 64 + q * 16; };
    gemm_i8<MT, 2, 4>(QD, aoff, pw.w + (wave * 2) * 4 * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 2 + j) * 16 + (lane & 15);
      int bias = pw.b[n], M = pw.M[n], sh = pw.sh[n];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int c8 = clamp8(requant(acc[i][j][e] + bias, M, sh, mode) + pw.zout);
          QP[(i * 16 + q * 4 + e) * QS + n] = (int8_t)lrelu_q(c8, lm);
        }
    }
  }
  __syncthreads();
  {
    i32x4 acc[MT][2];
    const int g = wave >> 1;
    auto aoff = [&](int i, int c) { return (i * 16 + m) * QS + g * 64 + q * 16; };
    gemm_i8<MT, 2, 1>(QP, aoff, cv.w + (wave * 2) * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 2 + j) * 16 + (lane & 15);
      int bias = cv.b[n], M = cv.M[n], sh = cv.sh[n];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int row = i * 16 + q * 4 + e;
          int c8 = clamp8(requant(acc[i][j][e] + bias, M, sh, mode) + cv.zout);
          QX[row * QS + n] = (int8_t)add_q(c8, (int)QX[row * QS + n], add);
        }
    }
  }
  __syncthreads();
}

}  // namespace lyra
