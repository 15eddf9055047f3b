// enc_kernels.hip -- SoundStream encoder (replaces soundstream_encoder.tflite as run by
// SoundStreamEncoder::Extract, lyra/soundstream_encoder.cc:53-64) as three stream-tiled gfx950 kernels.
//
//   enc_s0  S=8  streams/WG  PCM -> first conv k64/s16 -> 3 resblocks @64ch x 20 rows -> conv k10/s5   (fp32)
//   enc_s1  S=16 streams/WG  3 resblocks @128ch x 4 rows (2nd conv g=2) -> conv k4/s2 g=2               (fp32)
//   enc_s2  S=8  streams/WG  (enc_s2_kernel.hip) resblock @256 (fp32 dw+pw, then int8), 2 int8 resblocks,
//                            int8 k4/s2 g4, int8 k3 g4 bottleneck -> 64 int8 codes -> features
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

__global__ __launch_bounds__(NT0) void enc_s0_kernel(const EncS0P* __restrict__ Pp, const int16_t* __restrict__ pcm,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ out0) {
  const EncS0P& P = *Pp;
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

__global__ __launch_bounds__(NT1) void enc_s1_kernel(const EncS1P* __restrict__ Pp, const float* __restrict__ in0,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ out1) {
  const EncS1P& P = *Pp;
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

}  // namespace lyra
