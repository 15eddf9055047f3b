// resblocks_w.h -- the 64-channel residual-block triplet with WAVE-PRIVATE row tiles.
//
// resblocks64r (resblocks.h) splits a GEMM over the waves of a workgroup by OUTPUT CHANNEL: every wave needs every
// input channel, so the activations cross LDS between any two GEMMs and a block takes seven workgroup barriers.
// Here a wave owns ROWS: two 16-row M tiles (2 time steps x 16 streams) and all 64 channels of them, and the MFMA
// operands are swapped -- A = weight fragment (rows = output channels), B = activations (columns = the tile's rows):
//   D[4q + e][n] = sum_k W[perm(4q + e)][k] * act[row n][k],   lane = 16 q + n.
// With the weights' output channels permuted by AT16 inside every 16-tile (perm(4q + e) = 4e + q) the lane that holds
// row n gets, in register e of channel tile j, channel 16j + 4e + q -- bit for bit the B operand (k = 4 kk + q, kk = e) of
// the NEXT GEMM's K chunk j.  Depthwise out -> 1x1 GEMM -> LeakyReLU -> 1x1 GEMM -> residual add run register to
// register; LDS only carries lrelu(X) across rows for the dilated depthwise taps; a block has two (loose) barriers.
// Every dot product is still a k-ascending chain on v_mfma_f32_16x16x4_f32: results are bitwise those of resblocks64r.
#pragma once
#include "resblocks.h"

namespace lyra {

// X[u][j][e] = X[time 2 wave + u][stream lane & 15][channel 16 j + 4 e + (lane >> 4)]; rows (t, s) -> t * 16 + s, so
// that a tile is ONE time step of the 16 streams: which taps of the dilated depthwise conv fall before the frame is
// uniform over the wave (a scalar branch, no per-lane address-space select).
struct XTile { f32x4 v[2][4]; };

// acc[u][j] += W_j,c * B[u][c] over the four K chunks; weight fragments [(c * 4 + j) * 64 + lane] from L1 / L2.
// Rolling prefetch in 16 registers: fragment j of chunk c + 1 is requested right after the eight MFMAs that used
// fragment j of chunk c have issued (an MFMA reads its operands at issue), i.e. 24 MFMAs = 768 cycles before its
// first use.  Consecutive MFMAs alternate between the two tiles' accumulators (dependent distance 64 cycles).
__device__ __forceinline__ void gemm_sw(const f32x4* w_generic, const f32x4 (&b)[2][4], f32x4 (&acc)[2][4]) {
  const f32x4 LYRA_GLOBAL* wf = as_global(w_generic) + (threadIdx.x & 63);
  f32x4 f[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) f[j] = wf[j * 64];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j][kk], b[u][c][kk], acc[u][j], 0, 0, 0);
#ifdef WP_SAMEW   // timing experiment: every chunk re-reads chunk 0 (4 KB working set, certainly L1 hits)
      if (c < 3) f[j] = wf[j * 64 + (c + 1) * 0];
#else
      if (c < 3) f[j] = wf[((c + 1) * 4 + j) * 64];
#endif
    }
}

// A: LDS [20][16][CS] (rows t * 16 + s), shared by the ten waves of the workgroup.
template <int CS>
__device__ __forceinline__ void resblocks64w(XTile& X, float* A, const TileCtx& cx, const DwF* dws, const ConvF* pws,
                                             const ConvF* cvs, int off0, int off1, int off2) {
  constexpr int S = 16;
#pragma unroll 1
  for (int r = 0; r < 3; ++r) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // keep the index math inside the loop (see resblocks64r)
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int s = n;
    const int d = r == 0 ? 1 : (r == 1 ? 3 : 9);
    const int R2 = 2 * d;
    const int off = r == 0 ? off0 : (r == 1 ? off1 : off2);
    float* hist = reinterpret_cast<float*>(cx.sbase(s) + off);
    // 1. a = lrelu(X) of the wave's own rows -> A (AT16 order: channels 16j + 4e + q, e = 0..3, are contiguous)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = (2 * wave + u) * S + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&A[row * CS + 16 * j + 4 * q]) = lrelu4(X.v[u][j]);
    }
    __syncthreads();   // every row of a is in LDS
    // 2. depthwise k3, dilation d, directly in B-operand layout: taps inside the frame from A, older ones from the
    //    history (f32[R2][64], oldest first); the own row's tap comes from the registers
    f32x4 dv[2][4];
    {
      const float LYRA_GLOBAL* dww = as_global(dws[r].w) + 4 * q;
      const float LYRA_GLOBAL* dwb = as_global(dws[r].b) + 4 * q;
      const int wv = __builtin_amdgcn_readfirstlane(tid) >> 6;   // wave index as a scalar: the tap cases are branches
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * wv + u;
        const int r0 = t - 2 * d, r1 = t - d;       // tap rows (< 0: before the frame -> history), wave-uniform
        // one FLAT pointer per tap (an LDS row of A inside the frame, a history row in HBM before it): a single
        // flat_load per tap and chunk and straight-line code -- with separate LDS / global loads under (uniform)
        // branches the register allocator spilled over a hundred VGPRs in this phase
#ifdef WP_NOHIST   // timing experiment: taps before the frame read LDS row 0 instead of the history in HBM
        const float* p0 = &A[(max(r0, 0) * S + s) * CS + 4 * q];
        const float* p1 = &A[(max(r1, 0) * S + s) * CS + 4 * q];
#else
        const float* p0 = r0 >= 0 ? &A[(r0 * S + s) * CS + 4 * q] : hist + (R2 + r0) * 64 + 4 * q;
        const float* p1 = r1 >= 0 ? &A[(r1 * S + s) * CS + 4 * q] : hist + (R2 + r1) * 64 + 4 * q;
#endif
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(p0 + 16 * c);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(p1 + 16 * c);
          const f32x4 w0 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 16 * c);
          const f32x4 w1 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 64 + 16 * c);
          const f32x4 w2 = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dww + 128 + 16 * c);
          const f32x4 bb = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(dwb + 16 * c);
          f32x4 acc = fma4(v0, w0, (f32x4){0.f, 0.f, 0.f, 0.f});
          acc = fma4(v1, w1, acc);
          dv[u][c] = fma4(lrelu4(X.v[u][c]), w2, acc) + bb;
          asm volatile("" ::: "memory");   // one chunk's loads at a time (hoisting a tile's worth costs 112 VGPRs)
        }
      }
    }
    // 3. pointwise 64 -> 64, LeakyReLU; 4. 1x1 conv 64 -> 64 + residual -- register to register
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      gemm_sw(pws[r].w, dv, acc);
      const float LYRA_GLOBAL* b1 = as_global(pws[r].b) + 4 * q;   // AT16 order
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bj = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(b1 + 16 * j);
#pragma unroll
        for (int u = 0; u < 2; ++u) dv[u][j] = lrelu4(acc[u][j] + bj);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      gemm_sw(cvs[r].w, dv, acc);
      const float LYRA_GLOBAL* b2 = as_global(cvs[r].b) + 4 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bj = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(b2 + 16 * j);
#pragma unroll
        for (int u = 0; u < 2; ++u) X.v[u][j] = (acc[u][j] + bj) + X.v[u][j];
      }
    }
    __syncthreads();   // nobody reads this block's a (taps) or its history any more
    // 5. new history = the last R2 rows of a, each written by the wave that owns the row (from A, before that wave
    //    overwrites it in the next block)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * wave + u;
      if (t >= 20 - R2 && cx.valid(s)) {
        const int row = t * S + n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(hist + (t - (20 - R2)) * 64 + 16 * j + 4 * q) =
              *reinterpret_cast<const f32x4*>(&A[row * CS + 16 * j + 4 * q]);
      }
    }
  }
}

}  // namespace lyra
