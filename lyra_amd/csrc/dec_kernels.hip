// dec_kernels.hip -- the three decoder stages as kernels of their own (bodies: dec_stages.h).
#include "dec_stages.h"

#ifdef LYRA_TIMING
extern "C" int lyra_hip_debug_exit_at_d0(int i) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lyra_exit_at), &i, sizeof(int));
}
extern "C" int lyra_hip_debug_timing_d0(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_tdbg), sizeof(long long) * 128);
}
extern "C" int lyra_hip_debug_wgtrace_d0(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_wgtrace), sizeof(long long) * 2048 * 4);
}
#endif

#ifndef LYRA_I8_WAVES
#define LYRA_I8_WAVES 4   // waves per SIMD the int8 stage kernels are compiled for (5 -> at most 96 VGPRs)
#endif

#ifndef LYRA_C64_WAVES
#define LYRA_C64_WAVES 4   // waves per SIMD the 64-channel stage kernels are compiled for (3 -> up to 168 VGPRs, no spills)
#endif

namespace lyra {

size_t dec_s0_lds_bytes() { return dec_s0_lds(); }
int dec_s0_streams_per_wg() { return SD0; }

__global__ __launch_bounds__(NTD0, LYRA_I8_WAVES) void dec_s0_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                       const int32_t* __restrict__ ids, int B,
                                                       uint8_t* __restrict__ state, float* __restrict__ out0,
                                                       const uint8_t* __restrict__ packets, int num_stages,
                                                       const float* __restrict__ cb, int code_bytes, int tile0) {
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
  dec_s0_body<0>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}
__global__ __launch_bounds__(NTD0, LYRA_I8_WAVES) void dec_s0_dr_kernel(const DecS0P* __restrict__ Pp, const float* __restrict__ feats,
                                                          const int32_t* __restrict__ ids, int B,
                                                          uint8_t* __restrict__ state, float* __restrict__ out0,
                                                          const uint8_t* __restrict__ packets, int num_stages,
                                                          const float* __restrict__ cb, int code_bytes, int tile0) {
  if (((int)blockIdx.x + tile0) * SD0 >= B) return;
  dec_s0_body<1>(Pp, feats, ids, B, state, out0, packets, num_stages, cb, code_bytes, (int)blockIdx.x + tile0);
}

size_t dec_s1_lds_bytes() { return dec_s1_lds(); }
int dec_s1_streams_per_wg() { return SD1; }
int dec_s1_threads() { return NTD1; }

__global__ __launch_bounds__(NTD1, NTD1 == 512 ? 4 : 3) void dec_s1_kernel(const DecS1P* __restrict__ Pp, const float* __restrict__ in0,
                                                                          const int32_t* __restrict__ ids, int B,
                                                                          uint8_t* __restrict__ state, float* __restrict__ out1,
                                                                          int code_bytes, int tile0) {
  if (((int)blockIdx.x + tile0) * SD1 >= B) return;
  dec_s1_body(*Pp, in0, ids, B, state, out1, code_bytes, (int)blockIdx.x + tile0);
}

#ifdef LYRA_WAVE_PRIVATE
// =============================================================================================
// stage 2, wave-private residual blocks (resblocks_w.h): 16 streams per workgroup, ten waves, wave w owns time steps
// 2w and 2w+1 of the 16 streams (two 16-row M tiles) through all three blocks; the transposed conv at the end is the
// same GEMM as before on 24 M tiles.
// =============================================================================================
}  // namespace lyra
#include "resblocks_w.h"
namespace lyra {
namespace {
constexpr int SD2 = 16;
constexpr int NTD2 = 640;
}  // namespace

size_t dec_s2_lds_bytes() { return (size_t)(27 * SD2 * CS0 + SD2 * 48) * 4 + 64; }
int dec_s2_streams_per_wg() { return SD2; }
int dec_s2_threads() { return NTD2; }

__global__ __launch_bounds__(NTD2, 3) void dec_s2_kernel(const DecS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                          const int32_t* __restrict__ ids, int B,
                                                          uint8_t* __restrict__ state, int16_t* __restrict__ pcm,
                                                          int code_bytes, int tile0) {
  const DecS2P& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XB = smem;                     // [27][S][72]: rows 0-2 zeros, rows 3-22 activations, rows 23-26 zeros
  float* SB = XB + 27 * SD2 * CS0;      // old overlap tail [S][48]
  int* sids = reinterpret_cast<int*>(SB + SD2 * 48);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int b0 = ((int)blockIdx.x + tile0) * SD2;
  if (tid < SD2) sids[tid] = ids[min(b0 + tid, B - 1)];
  const auto warm = l2_warm<NTD2, 1>(P.warm);
  const auto warm_code = code_warm<NTD2>(code_bytes);
  __syncthreads();
  TileCtx cx{state, sids, nullptr, B - b0, st::D2_BYTES};   // T = 20 >= every 2*dilation: no ring, no phase
  XTile X;   // residual stream, this wave's 32 rows x 64 channels
  {
    const int s = m, b = min(b0 + s, B - 1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * wave + u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        X.v[u][j] = *reinterpret_cast<const f32x4*>(&in1[((size_t)b * 20 + t) * 64 + 16 * j + 4 * q]);
    }
  }
  for (int idx = tid; idx < 7 * SD2 * 16; idx += NTD2) {
    int p4 = idx & 15, s = (idx >> 4) & (SD2 - 1), j = (idx >> 4) / SD2;
    int row = j < 3 ? j : 20 + j;
    *reinterpret_cast<f32x4*>(&XB[(row * SD2 + s) * CS0 + p4 * 4]) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int idx = tid; idx < SD2 * 48; idx += NTD2) {
    int s = idx / 48, i = idx - s * 48;
    SB[idx] = reinterpret_cast<const float*>(cx.sbase(s) + st::D_UP3)[i];
  }
  float* A = XB + 3 * SD2 * CS0;
  resblocks64w<CS0>(X, A, cx, P.dw, P.pw, P.cv, st::D_R2_0, st::D_R2_1, st::D_R2_2);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (2 * wave + u) * SD2 + m;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&A[row * CS0 + 16 * j + 4 * q]) = lrelu4(X.v[u][j]);
  }
  __syncthreads();
  // tconv k64/s16, polyphase: blocks b = 0..22 (+1 of padding), rows (b, s); K = 4 x 64 (oldest input first);
  // N = 16 phases.  24 * 16 rows = 24 M tiles: waves 0..7 take three each.
  if (wave < 8) {
    f32x4 acc[3][1];
    auto aoff = [&](int i, int c) {
      int R = (3 * wave + i) * 16 + m, b = R / SD2, s = R & (SD2 - 1);
      return ((b + (c >> 2)) * SD2 + s) * CS0 + (c & 3) * 16 + q * 4;
    };
    gemm_f32<3, 1, 16>(XB, aoff, P.up.w, acc);
    const int j = lane & 15;
    const float bias = as_global(P.up.b)[0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int R = (3 * wave + i) * 16 + q * 4 + e, b = R / SD2, s = R & (SD2 - 1);
        if (b > 22) continue;
        const int tau = 16 * b + j;
        float y = acc[i][0][e] + bias;
        y = y + (tau < 48 ? SB[s * 48 + tau] : 0.f);
        if (!cx.valid(s)) continue;
        if (tau < 320) {
          // UnitToInt16Scalar (dsp_utils.h:54-88): scale, clip, C truncation
          float v = y * 32768.f;
          v = v < -32768.f ? -32768.f : v;
          v = v > 32767.f ? 32767.f : v;
          pcm[(size_t)(b0 + s) * 320 + tau] = (int16_t)v;
        } else {
          reinterpret_cast<float*>(cx.sbase(s) + st::D_UP3)[tau - 320] = y - P.up_sub;
        }
      }
  }
  l2_warm_sink(warm, state, B);
  l2_warm_sink(warm_code, state, B);
}
#else
namespace {
#ifndef LYRA_S0_STREAMS
#define LYRA_S0_STREAMS 4
#endif
constexpr int SD2 = LYRA_S0_STREAMS;   // 4 streams with 256 threads, or 8 with 512
}  // namespace

size_t dec_s2_lds_bytes() { return dec_s2_lds(SD2); }
int dec_s2_streams_per_wg() { return SD2; }
int dec_s2_threads() { return 64 * SD2; }

__global__ __launch_bounds__(64 * SD2, LYRA_C64_WAVES) void dec_s2_kernel(const DecS2P* __restrict__ Pp, const float* __restrict__ in1,
                                                            const int32_t* __restrict__ ids, int B,
                                                            uint8_t* __restrict__ state, int16_t* __restrict__ pcm,
                                                            int code_bytes, int tile0) {
  if (((int)blockIdx.x + tile0) * SD2 >= B) return;
  dec_s2_body<SD2>(*Pp, in1, ids, B, state, pcm, code_bytes, (int)blockIdx.x + tile0);
}
#endif  // LYRA_WAVE_PRIVATE

}  // namespace lyra
