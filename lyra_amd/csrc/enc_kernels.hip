// enc_kernels.hip -- SoundStream encoder (replaces soundstream_encoder.tflite as run by
// SoundStreamEncoder::Extract, lyra/soundstream_encoder.cc:53-64) as three stream-tiled gfx950 kernels.
//
//   enc_s0  S=8  streams/WG  PCM -> first conv k64/s16 -> 3 resblocks @64ch x 20 rows -> conv k10/s5   (fp32)
//   enc_s1  S=16 streams/WG  3 resblocks @128ch x 4 rows (2nd conv g=2) -> conv k4/s2 g=2               (fp32)
//   enc_s2  S=16 streams/WG  resblock @256 (fp32 dw+pw, then int8), 2 int8 resblocks, int8 k4/s2 g4,
//                            int8 k3 g4 bottleneck -> 64 int8 codes -> features
//
// Every fp32 dot product runs on v_mfma_f32_16x16x4_f32 in ascending-k order (== the oracle's fmaf chain),
// int8 ones on v_mfma_i32_16x16x64_i8; everything between two GEMMs (LeakyReLU, depthwise dilated conv,
// residual add, (de)quantisation, state update) is fused around them in LDS/registers.  Per stream and
// step the only HBM traffic is: PCM in, state read/write, two small inter-stage activations, features out.
#include "resblocks.h"

namespace lyra {

// =============================================================================================
// stage 0
// =============================================================================================
namespace {
constexpr int S0 = 8;      // streams per workgroup
constexpr int CS0 = 72;    // LDS row stride (64 + 8) floats
constexpr int PBS = 376;   // PCM staging row stride (368 + 8) floats
constexpr int NT0 = 512;   // threads
}  // namespace

size_t enc_s0_lds_bytes() { return (size_t)(25 * S0 * CS0 + 20 * S0 * CS0) * 4 + 64; }
int enc_s0_streams_per_wg() { return S0; }

__global__ __launch_bounds__(NT0) void enc_s0_kernel(EncS0P P, const int16_t* __restrict__ pcm,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ out0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                     // [25][S0][CS0]: rows 0-4 strided-conv history, rows 5-24 X[t]
  float* DB = XB + 25 * S0 * CS0;       // [20][S0][CS0]: depthwise out / pointwise out; first: PCM staging
  int* sids = reinterpret_cast<int*>(DB + 20 * S0 * CS0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * S0;
  if (tid < S0) sids[tid] = ids[min(b0 + tid, B - 1)];
  __syncthreads();
  auto sbase = [&](int s) -> uint8_t* { return state + (size_t)sids[s] * st::BYTES; };
  auto valid = [&](int s) -> bool { return b0 + s < B; };

  // ---- A. window = [48 history samples | 320 new samples] / 32768, AT16 order ----------------
  float* PB = DB;
  for (int idx = tid; idx < S0 * 368; idx += NT0) {
    int s = idx / 368, i = idx - s * 368;
    float v;
    if (i < 48) {
      v = reinterpret_cast<const float*>(sbase(s) + st::E_FIRST)[i];
    } else {
      int b = min(b0 + s, B - 1);
      v = (float)pcm[(size_t)b * 320 + (i - 48)] * (1.0f / 32768.0f);  // Int16ToUnitScalar, dsp_utils.h:106-108
    }
    PB[s * PBS + at16(i)] = v;
  }
  __syncthreads();
  for (int idx = tid; idx < S0 * 48; idx += NT0) {
    int s = idx / 48, i = idx - s * 48;
    if (valid(s)) reinterpret_cast<float*>(sbase(s) + st::E_FIRST)[i] = PB[s * PBS + at16(320 + i)];
  }

  const int wn = wave & 3, wm = wave >> 2;  // GEMM wave grid for N=64: 4 N tiles x 2 halves of the 10 M tiles
  const int ncol = wn * 16 + (lane & 15);   // logical output channel of this lane's C column
  const int pcol = at16(ncol);

  // ---- B. first conv k64/s16: [20x8 rows] x K=64 x N=64 ---------------------------------------
  {
    f32x4 acc[5][1];
    auto aoff = [&](int i, int c) {
      int t = 2 * (wm * 5 + i) + (m >> 3);
      return (m & 7) * PBS + (t + c) * 16 + q * 4;
    };
    gemm_f32<5, 1, 4>(PB, aoff, P.first.w + wn * 4 * 64, acc);
    float bias = P.first.b[ncol];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) XB[(40 + (wm * 5 + i) * 16 + q * 4 + e) * CS0 + pcol] = acc[i][0][e] + bias;
  }
  __syncthreads();

  // ---- C. three residual blocks, dilation 1 / 3 / 9 --------------------------------------------
  TileCtx cx{state, sids, nullptr, B - b0};
  resblocks64(XB + 5 * S0 * CS0, DB, cx, P.dw, P.pw, P.cv, st::E_R0_0, st::E_R0_1, st::E_R0_2);

  // ---- D. a = lrelu(X); prepend the 5 history rows of the strided conv ---------------------------
  for (int idx = tid; idx < 20 * S0 * 16; idx += NT0) {
    int p4 = idx & 15, rs = idx >> 4;
    f32x4* x = reinterpret_cast<f32x4*>(&XB[(40 + rs) * CS0 + p4 * 4]);
    *x = lrelu4(*x);
  }
  for (int idx = tid; idx < 5 * S0 * 16; idx += NT0) {
    int p4 = idx & 15, s = (idx >> 4) & 7, j = idx >> 7;
    *reinterpret_cast<f32x4*>(&XB[(j * S0 + s) * CS0 + p4 * 4]) =
        *reinterpret_cast<const f32x4*>(sbase(s) + st::E_D0 + (j * 64 + p4 * 4) * 4);
  }
  __syncthreads();
  for (int idx = tid; idx < 5 * S0 * 16; idx += NT0) {
    int p4 = idx & 15, s = (idx >> 4) & 7, j = idx >> 7;
    if (valid(s))
      *reinterpret_cast<f32x4*>(sbase(s) + st::E_D0 + (j * 64 + p4 * 4) * 4) =
          *reinterpret_cast<const f32x4*>(&XB[((20 + j) * S0 + s) * CS0 + p4 * 4]);
  }

  // ---- E. conv k10/s5: [4x8 rows] x K=640 x N=128 -----------------------------------------------
  {
    f32x4 acc[2][1];
    auto aoff = [&](int i, int c) {
      int tap = c >> 2, c16 = c & 3;
      int tau = 2 * i + (m >> 3);
      return ((5 * tau + tap) * S0 + (m & 7)) * CS0 + c16 * 16 + q * 4;
    };
    gemm_f32<2, 1, 40>(XB, aoff, P.down.w + wave * 40 * 64, acc);
    int n = wave * 16 + (lane & 15);
    float bias = P.down.b[n];
    int pc = at16(n);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int rr = q * 4 + e, tau = 2 * i + (rr >> 3), s = rr & 7;
        if (valid(s)) out0[((size_t)(b0 + s) * 4 + tau) * 128 + pc] = acc[i][0][e] + bias;
      }
  }
}

// =============================================================================================
// stage 1
// =============================================================================================
namespace {
constexpr int S1 = 16;
constexpr int CS1 = 136;   // 128 + 8
constexpr int NT1 = 512;
}  // namespace

size_t enc_s1_lds_bytes() { return (size_t)(6 * S1 * CS1 + 4 * S1 * CS1) * 4 + 2 * S1 * 4; }
int enc_s1_streams_per_wg() { return S1; }

__global__ __launch_bounds__(NT1) void enc_s1_kernel(EncS1P P, const float* __restrict__ in0,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                    // [6][S1][CS1]: rows 0-1 strided-conv history, rows 2-5 X[t]
  float* DB = XB + 6 * S1 * CS1;       // [4][S1][CS1]
  int* sids = reinterpret_cast<int*>(DB + 4 * S1 * CS1);
  int* sphase = sids + S1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * S1;
  if (tid < S1) {
    int id = ids[min(b0 + tid, B - 1)];
    sids[tid] = id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)id * st::BYTES + st::ENC_PHASE);
  }
  __syncthreads();
  auto sbase = [&](int s) -> uint8_t* { return state + (size_t)sids[s] * st::BYTES; };
  auto valid = [&](int s) -> bool { return b0 + s < B; };

  for (int idx = tid; idx < 4 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, s = (idx >> 5) & 15, t = idx >> 9;
    int b = min(b0 + s, B - 1);
    *reinterpret_cast<f32x4*>(&XB[((2 + t) * S1 + s) * CS1 + p4 * 4]) =
        *reinterpret_cast<const f32x4*>(&in0[((size_t)b * 4 + t) * 128 + p4 * 4]);
  }
  __syncthreads();

  TileCtx cx{state, sids, sphase, B - b0};
  resblocks128(XB + 2 * S1 * CS1, DB, cx, P.dw, P.pw, P.cv, st::E_R1_0, st::E_R1_1, st::E_R1_2);

  for (int idx = tid; idx < 4 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, rs = idx >> 5;
    f32x4* x = reinterpret_cast<f32x4*>(&XB[(2 * S1 + rs) * CS1 + p4 * 4]);
    *x = lrelu4(*x);
  }
  for (int idx = tid; idx < 2 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, s = (idx >> 5) & 15, j = idx >> 9;
    *reinterpret_cast<f32x4*>(&XB[(j * S1 + s) * CS1 + p4 * 4]) =
        *reinterpret_cast<const f32x4*>(sbase(s) + st::E_D1 + (j * 128 + p4 * 4) * 4);
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S1 * 32; idx += NT1) {
    int p4 = idx & 31, s = (idx >> 5) & 15, j = idx >> 9;
    if (valid(s))
      *reinterpret_cast<f32x4*>(sbase(s) + st::E_D1 + (j * 128 + p4 * 4) * 4) =
          *reinterpret_cast<const f32x4*>(&XB[((4 + j) * S1 + s) * CS1 + p4 * 4]);
  }

  {  // conv k4/s2, 2 groups: per group [2x16 rows] x K=256 x N=128
    f32x4 acc[2][2];
    const int g = wave >> 2, nt0 = g * 8 + (wave & 3) * 2;
    auto aoff = [&](int i, int c) {
      int tap = c >> 2, c16 = c & 3;
      return ((2 * i + tap) * S1 + m) * CS1 + g * 64 + c16 * 16 + q * 4;
    };
    gemm_f32<2, 2, 16>(XB, aoff, P.down.w + nt0 * 16 * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (nt0 + j) * 16 + (lane & 15);
      float bias = P.down.b[n];
      int pc = at16(n);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int s = q * 4 + e;
          if (valid(s)) out1[((size_t)(b0 + s) * 2 + i) * 256 + pc] = acc[i][j][e] + bias;
        }
    }
  }
}

// =============================================================================================
// stage 2
// =============================================================================================
namespace {
constexpr int S2 = 16;
constexpr int CS2 = 264;   // 256 + 8 floats
constexpr int QS = 288;    // int8 row stride, C = 256
constexpr int QS5 = 544;   // int8 row stride, C = 512
constexpr int NT2 = 512;
constexpr int XF_BYTES = 2 * S2 * CS2 * 4;  // 33,792
constexpr int QB_BYTES = 2 * S2 * QS;       // 9,216
}  // namespace

size_t enc_s2_lds_bytes() { return (size_t)2 * XF_BYTES + 4 * QB_BYTES + 2 * S2 * 4; }
int enc_s2_streams_per_wg() { return S2; }

__global__ __launch_bounds__(NT2) void enc_s2_kernel(EncS2P P, const float* __restrict__ in1,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ feats,
                                                      float* __restrict__ codes_dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XF = smem;                                  // [2][S2][CS2] stage input (skip of resblock 0)
  float* DF = XF + 2 * S2 * CS2;                     // [2][S2][CS2] depthwise out
  int8_t* QX = reinterpret_cast<int8_t*>(DF + 2 * S2 * CS2);  // residual stream (int8)
  int8_t* QA = QX + QB_BYTES;
  int8_t* QD = QA + QB_BYTES;
  int8_t* QP = QD + QB_BYTES;
  int* sids = reinterpret_cast<int*>(QP + QB_BYTES);
  int* sphase = sids + S2;
  int8_t* QB4 = reinterpret_cast<int8_t*>(DF);       // [4][S2][QS]   (aliases DF once it is dead)
  int8_t* QC = reinterpret_cast<int8_t*>(XF);        // [3][S2][QS5]  (aliases XF once it is dead)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = blockIdx.x * S2;
  const int mode = P.mode;
  if (tid < S2) {
    int id = ids[min(b0 + tid, B - 1)];
    sids[tid] = id;
    sphase[tid] = *reinterpret_cast<const int*>(state + (size_t)id * st::BYTES + st::ENC_PHASE);
  }
  __syncthreads();
  auto sbase = [&](int s) -> uint8_t* { return state + (size_t)sids[s] * st::BYTES; };
  auto valid = [&](int s) -> bool { return b0 + s < B; };

  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int p4 = idx & 63, s = (idx >> 6) & 15, t = idx >> 10;
    int b = min(b0 + s, B - 1);
    *reinterpret_cast<f32x4*>(&XF[(t * S2 + s) * CS2 + p4 * 4]) =
        *reinterpret_cast<const f32x4*>(&in1[((size_t)b * 2 + t) * 256 + p4 * 4]);
  }
  __syncthreads();

  // ---- resblock 0, fp32 half: depthwise (dil 1, history 2 rows, replaced) + pointwise 256->256 ----
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int p4 = idx & 63, s = (idx >> 6) & 15, t = idx >> 10;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int tau = t - (2 - j);
      f32x4 v;
      if (tau >= 0) v = lrelu4(*reinterpret_cast<const f32x4*>(&XF[(tau * S2 + s) * CS2 + p4 * 4]));
      else v = *reinterpret_cast<const f32x4*>(sbase(s) + st::E_R2_0 + ((2 + tau) * 256 + p4 * 4) * 4);
      acc = fma4(v, *reinterpret_cast<const f32x4*>(&P.dw0.w[j * 256 + p4 * 4]), acc);
    }
    f32x4 bb = *reinterpret_cast<const f32x4*>(&P.dw0.b[p4 * 4]);
    *reinterpret_cast<f32x4*>(&DF[(t * S2 + s) * CS2 + p4 * 4]) = acc + bb;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int p4 = idx & 63, s = (idx >> 6) & 15, j = idx >> 10;
    if (valid(s))
      *reinterpret_cast<f32x4*>(sbase(s) + st::E_R2_0 + (j * 256 + p4 * 4) * 4) =
          lrelu4(*reinterpret_cast<const f32x4*>(&XF[(j * S2 + s) * CS2 + p4 * 4]));
  }
  {  // pointwise fp32 -> QUANTIZE -> int8 LeakyReLU -> QP
    f32x4 acc[2][2];
    auto aoff = [&](int i, int c) { return (i * 16 + m) * CS2 + c * 16 + q * 4; };
    gemm_f32<2, 2, 16>(DF, aoff, P.pw0.w + (wave * 2) * 16 * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 2 + j) * 16 + (lane & 15);
      float bias = P.pw0.b[n];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int q8 = quantize_f(acc[i][j][e] + bias, P.q_r0.s, P.q_r0.z);
          QP[(i * 16 + q * 4 + e) * QS + n] = (int8_t)lrelu_q(q8, P.lr[0]);
        }
    }
  }
  __syncthreads();
  {  // grouped 1x1 int8 (4 groups 64->64) -> DEQUANTIZE + float skip -> QUANTIZE = X1
    i32x4 acc[2][2];
    const int g = wave >> 1;
    auto aoff = [&](int i, int c) { return (i * 16 + m) * QS + g * 64 + q * 16; };
    gemm_i8<2, 2, 1>(QP, aoff, P.r0b.w + (wave * 2) * 64, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = (wave * 2 + j) * 16 + (lane & 15);
      int bias = P.r0b.b[n], M = P.r0b.M[n], sh = P.r0b.sh[n];
      int pc = at16(n);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int row = i * 16 + q * 4 + e;
          int c8 = clamp8(requant(acc[i][j][e] + bias, M, sh, mode) + P.r0b.zout);
          float v = dequantize_f(c8, P.dq_r0.s, P.dq_r0.z) + XF[row * CS2 + pc];
          QX[row * QS + n] = (int8_t)quantize_f(v, P.q_x1.s, P.q_x1.z);
        }
    }
  }
  __syncthreads();

  // ---- int8 resblocks 1, 2 (dilation 3 / 9; ring histories of 6 / 18 rows, T = 2) ---------------
  TileCtx cx{state, sids, sphase, B - b0};
  resblock_q256(QX, QA, QD, QP, cx, 3, st::E_R2_1, P.lr[1], P.lr[2], P.dwq[0], P.pwq[0], P.cvq[0], P.add[0], mode);
  resblock_q256(QX, QA, QD, QP, cx, 9, st::E_R2_2, P.lr[3], P.lr[4], P.dwq[1], P.pwq[1], P.cvq[1], P.add[1], mode);

  // ---- int8 LeakyReLU, 2-row history (replaced), conv k4/s2 g4 -> [1][512] --------------------------
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int w4 = idx & 63, s = (idx >> 6) & 15, t = idx >> 10;
    int w = *reinterpret_cast<const int*>(&QX[(t * S2 + s) * QS + w4 * 4]);
    *reinterpret_cast<int*>(&QB4[((2 + t) * S2 + s) * QS + w4 * 4]) =
        pack8(lrelu_q(sx8(w, 0), P.lr[5]), lrelu_q(sx8(w, 1), P.lr[5]), lrelu_q(sx8(w, 2), P.lr[5]),
              lrelu_q(sx8(w, 3), P.lr[5]));
    *reinterpret_cast<int*>(&QB4[(t * S2 + s) * QS + w4 * 4]) =
        *reinterpret_cast<const int*>(sbase(s) + st::E_D2 + t * 256 + w4 * 4);
  }
  // bottleneck history (ring R=2, T=1): rows [f-2, f-1] -> QC rows 0, 1
  for (int idx = tid; idx < 2 * S2 * 128; idx += NT2) {
    int w4 = idx & 127, s = (idx >> 7) & 15, j = idx >> 11;
    int slot = (sphase[s] + j) & 1;
    *reinterpret_cast<int*>(&QC[(j * S2 + s) * QS5 + w4 * 4]) =
        *reinterpret_cast<const int*>(sbase(s) + st::E_BOTT + slot * 512 + w4 * 4);
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S2 * 64; idx += NT2) {
    int w4 = idx & 63, s = (idx >> 6) & 15, t = idx >> 10;
    if (valid(s))
      *reinterpret_cast<int*>(sbase(s) + st::E_D2 + t * 256 + w4 * 4) =
          *reinterpret_cast<const int*>(&QB4[((2 + t) * S2 + s) * QS + w4 * 4]);
  }
  {
    i32x4 acc[1][4];
    const int g = wave >> 1;
    auto aoff = [&](int i, int c) { return (c * S2 + m) * QS + g * 64 + q * 16; };
    gemm_i8<1, 4, 4>(QB4, aoff, P.down2.w + (wave * 4) * 4 * 64, acc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = (wave * 4 + j) * 16 + (lane & 15);
      int bias = P.down2.b[n], M = P.down2.M[n], sh = P.down2.sh[n];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int s = q * 4 + e;
        int c8 = clamp8(requant(acc[0][j][e] + bias, M, sh, mode) + P.down2.zout);
        QC[(2 * S2 + s) * QS5 + n] = (int8_t)lrelu_q(c8, P.lr[6]);
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < S2 * 128; idx += NT2) {
    int w4 = idx & 127, s = idx >> 7;
    int slot = sphase[s] & 1;
    if (valid(s))
      *reinterpret_cast<int*>(sbase(s) + st::E_BOTT + slot * 512 + w4 * 4) =
          *reinterpret_cast<const int*>(&QC[(2 * S2 + s) * QS5 + w4 * 4]);
  }
  // ---- bottleneck conv k3 g4: per group K = 3*128, N = 16 -> 64 int8 codes ----------------------------
  if (wave < 4) {
    i32x4 acc[1][1];
    const int g = wave;
    auto aoff = [&](int i, int c) { return ((c >> 1) * S2 + m) * QS5 + g * 128 + (c & 1) * 64 + q * 16; };
    gemm_i8<1, 1, 6>(QC, aoff, P.bott.w + g * 6 * 64, acc);
    int n = g * 16 + (lane & 15);
    int bias = P.bott.b[n], M = P.bott.M[n], sh = P.bott.sh[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = q * 4 + e;
      int c8 = clamp8(requant(acc[0][0][e] + bias, M, sh, mode) + P.bott.zout);
      if (valid(s)) {
        feats[(size_t)(b0 + s) * 64 + n] = dequantize_f(c8, P.out.s, P.out.z);
        if (codes_dbg) codes_dbg[(size_t)(b0 + s) * 64 + n] = (float)c8;
      }
    }
  }
  if (tid < S2 && valid(tid)) {
    int ph = sphase[tid] + 1;
    *reinterpret_cast<int*>(sbase(tid) + st::ENC_PHASE) = ph >= st::PHASE_MOD ? 0 : ph;
  }
}

}  // namespace lyra
