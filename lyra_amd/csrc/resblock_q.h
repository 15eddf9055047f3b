// resblock_q.h -- the int8 residual block at 256 channels x 2 rows shared by encoder stage 2 and decoder
// stage 0 (graph ops 107-133 of soundstream_encoder.tflite / 93-117 of lyragan.tflite).
// QX residual stream, QA/QD/QP scratch, all [2][S][288] int8.  Ring history of R2 = 2*d rows, T = 2.
#pragma once
#include "resblocks.h"

namespace lyra {

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
  const int tb = d == 3 ? 20 : 30;
  (void)tb;
  LYRA_TSTAMP(tb + 0);
  for (int idx = tid; idx < 2 * S * 64; idx += NT) {
    int w4 = idx & 63, rs = idx >> 6;
    int w = *reinterpret_cast<const int*>(&QX[rs * QS + w4 * 4]);
    *reinterpret_cast<int*>(&QA[rs * QS + w4 * 4]) =
        pack8(lrelu_q(sx8(w, 0), la), lrelu_q(sx8(w, 1), la), lrelu_q(sx8(w, 2), la), lrelu_q(sx8(w, 3), la));
  }
  __syncthreads();
  LYRA_TSTAMP(tb + 1);
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
  LYRA_TSTAMP(tb + 2);
  for (int idx = tid; idx < 2 * S * 64; idx += NT) {
    int w4 = idx & 63, s = (idx >> 6) & (S - 1), t = (idx >> 6) / S;
    int row = (cx.sphase[s] * 2) % R2 + t;
    row = row >= R2 ? row - R2 : row;
    if (cx.valid(s))
      *reinterpret_cast<int*>(cx.sbase(s) + off + row * 256 + w4 * 4) =
          *reinterpret_cast<const int*>(&QA[(t * S + s) * QS + w4 * 4]);
  }
  LYRA_TSTAMP(tb + 3);
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
  LYRA_TSTAMP(tb + 4);
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
