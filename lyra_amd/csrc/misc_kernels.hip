// misc_kernels.hip -- residual vector quantizer, packet (un)packing, log-mel front end, state reset.
#include "kernels.h"

#ifdef LYRA_TIMING
extern "C" int lyra_hip_debug_timing_misc(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lyra_tdbg), sizeof(long long) * 128);
}
#endif

namespace lyra {

#ifdef LYRA_PARKED
// The 64-term distance chain of one codeword, dims in ascending order: df = r - c, sq = df * df, sum = sum + sq, three
// separate fp32 operations per term as the graph's SUB / MUL / SUM (residual_vector_quantizer.cc:77-110 runs them
// through the `encode` subgraph).  `mine` = the lane's four residual dims; lane d4 of the row holds dims 4*d4..4*d4+3.
// x(lane L of the 16-lane row) - c: the row broadcast (DPP row_newbcast) rides on the subtraction's first operand, so the
// residual never travels through LDS and costs no instruction of its own.  (The compiler's DPP combiner does not fold a
// row_newbcast v_mov into its user, hence the instruction is spelled out; x must not have been written by the two
// preceding VALU instructions -- stage() separates the update of `mine` from the next chain.)
template <int L>
__device__ __forceinline__ float bcast_sub(float x, float c) {
  float d;
  asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(c), "n"(L));
  return d;
}
template <int D4>
__device__ __forceinline__ void rvq_terms(float& sum, float m0, float m1, float m2, float m3, const f32x4 (&row)[16]) {
  if constexpr (D4 < 16) {
    const f32x4 c = row[D4];
    const f32x4 df = {bcast_sub<D4>(m0, c[0]), bcast_sub<D4>(m1, c[1]), bcast_sub<D4>(m2, c[2]), bcast_sub<D4>(m3, c[3])};
    const f32x4 sq = df * df;
    sum = sum + sq[0];
    sum = sum + sq[1];
    sum = sum + sq[2];
    sum = sum + sq[3];
    rvq_terms<D4 + 1>(sum, m0, m1, m2, m3, row);
  }
}

// =============================================================================================
// RVQ encode, ALL-EXACT CHAIN FORM (rounds 1-3; since round 4 only in the variant build, for A/B and as the witness the
// shipped kernel is tested against, tests/test_gpu_side_kernels.py): quantizer.tflite `encode` (555 ops) + the bit-string
// assembly of ResidualVectorQuantizer::Quantize (lyra/residual_vector_quantizer.cc:77-110) + Packet<>::Pack
// (lyra/packet.h:91-122).  16 lanes = the 16 codewords of a stage; each lane runs the 64-term
// squared-distance sum in the oracle's order (separate multiply and add, d ascending), then a
// 16-lane DPP argmin with lowest-index tie break (ARG_MIN = first minimum).  4 frames per wavefront,
// 16 per workgroup.
//
// With 4096 frames there is exactly one 64-term chain per lane of the chip and one wavefront per SIMD: the kernel
// is a 46-step dependent chain and nothing but its own latency matters.  Per stage the critical path is
//   residual broadcast reads (LDS) -> 64 dependent adds -> 4 DPP steps -> winner row read (LDS) -> 3 ops -> LDS write,
// so everything that does not depend on the residual is taken off it:
//  * the lane's codeword row of stage k+1 (16 x ds_read_b128) is fetched into registers while stage k reduces and
//    updates (two register sets, the stage loop is unrolled by two);
//  * the residual lives in LDS (rs, row stride 68 floats so the four frames of a wavefront hit different banks),
//    shared by the 16 lanes of its frame with broadcast reads; lane j owns dims 4j..4j+3, keeps them in registers
//    and applies the graph's three fp32 ops r - (r + (q - r)) to them once per stage.  All 16 lanes of a frame sit
//    in one wavefront and LDS operations of a wavefront execute in order: the write-back needs no barrier;
//  * codebooks go global -> registers -> LDS in windows of W = 8 stages (rows padded to 68 floats: conflict-free
//    ds_read_b128 across the 16 code lanes), three buffers deep, one barrier per window: window w+1 is already in
//    LDS while window w computes (the cross-window prefetch needs it), window w+2 is in flight from L2.
// =============================================================================================
// W: stages per codebook window; PREFETCH: fetch the next stage's codeword row into a second register set while the
// current stage reduces (244 VGPRs) or read it at the start of its own stage (<= 128 VGPRs).
template <int W, bool PREFETCH>
__device__ __forceinline__ void rvq_encode_body(const float* __restrict__ cb, const float* __restrict__ feats, int B,
                                                int num_stages, int32_t* __restrict__ indices,
                                                uint8_t* __restrict__ packets, const int32_t* __restrict__ mask_ids,
                                                int32_t* __restrict__ packet_bytes) {
  constexpr int ROW = 68, WFLOATS = W * 16 * ROW;
  __shared__ __attribute__((aligned(16))) float cbs[3][WFLOATS];
  const int tid = threadIdx.x;
  const int j = tid & 15;
  const int frame = blockIdx.x * 16 + (tid >> 4);
  const int f = min(frame, B - 1);
  // DTX (lyra_encoder.cc:136-141): a stream whose hop is noise gets an empty packet; mask_ids[frame] < 0 marks it
  const bool live = frame < B && !(mask_ids && mask_ids[f] < 0);
  const int ldrow = tid >> 4, ldc4 = tid & 15;  // this thread's float4 of each stage's [16][64] codebook
  const f32x4 LYRA_GLOBAL* cbg = reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(cb)) + ldrow * 16 + ldc4;
  f32x4 stage_in[W];
  auto gload = [&](int win) {
#pragma unroll
    for (int v = 0; v < W; ++v) stage_in[v] = cbg[(size_t)min(win * W + v, 45) * 256];
  };   // (windows past the last stage re-read stage 45: never used, keeps the loop free of tail cases)
  auto lstore = [&](int win) {
    float* dst = cbs[win % 3] + ldrow * ROW + ldc4 * 4;
#pragma unroll
    for (int v = 0; v < W; ++v) *reinterpret_cast<f32x4*>(dst + v * 16 * ROW) = stage_in[v];
  };
  gload(0);
  f32x4 mine = *reinterpret_cast<const f32x4*>(&feats[(size_t)f * 64 + j * 4]);   // this lane's four residual dims
  lstore(0);
  gload(1);
  __syncthreads();
  const int nbytes = (num_stages + 1) >> 1;
  int cur = 0;
  f32x4 rowa[16], rowb[16];
  auto load_row = [&](f32x4 (&row)[16], int k) {
    const float* c = cbs[(k / W) % 3] + (k & (W - 1)) * 16 * ROW + j * ROW;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) row[d4] = *reinterpret_cast<const f32x4*>(&c[d4 * 4]);
  };
  if (PREFETCH) load_row(rowa, 0);
  auto stage = [&](int k, const f32x4 (&row)[16], f32x4 (&next)[16]) {
    const int u = k & (W - 1), win = k / W;
    if (u == 0) {
      // window win+1 -> LDS (fetched a window ago), request window win+2.  Buffer (win+1) % 3 last held window
      // win-2, which nobody reads any more: every wave passed the previous window's barrier, i.e. finished
      // window win-2, before any wave could get here.
      lstore(win + 1);
      gload(win + 2);
      __syncthreads();
    }
    if (!PREFETCH) load_row(next, k);   // (`row` and `next` are the same register set then)
    // residual dims 4*d4 .. 4*d4+3 live in lane d4 of the frame's 16-lane row: DPP row broadcast (row_newbcast, folded
    // into the subtraction's operand fetch) instead of a round trip through LDS
    float sum = 0.f;
    {
      float m0 = mine[0], m1 = mine[1], m2 = mine[2], m3 = mine[3];
      // a DPP operand must not have been written by the two preceding VALU instructions (`mine` is updated at the end
      // of the previous stage); tied to the data so it cannot be scheduled away from between the two
      asm volatile("s_nop 1" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
      rvq_terms<0>(sum, m0, m1, m2, m3, PREFETCH ? row : next);
    }
    // Off the critical path: issued after the chain and before the reduction, so the 16 reads drain while the DPP
    // steps run and the winner-row read below finds the LDS idle.
    __builtin_amdgcn_sched_barrier(0);
    if (PREFETCH && k + 1 < num_stages) load_row(next, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ARG_MIN = first minimum, branch-free: the row minimum of the distance by a 16-lane all-reduce with DPP row
    // rotations (register-only, no LDS crossbar), then the lowest lane of the frame's 16-lane field that holds it
    // (ballot + find-first-set).  Distances are sums of squares (>= +0, finite for finite features).
    // (as unsigned integers: non-negative floats order like their bit patterns -- one v_min_u32 with a DPP operand per
    // step instead of move + two canonicalising maxima + minimum)
    unsigned mbits = __builtin_bit_cast(unsigned, sum);
#define LYRA_ROR_MINU(N) \
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(mbits) : "v"(mbits));
    LYRA_ROR_MINU(8) LYRA_ROR_MINU(4) LYRA_ROR_MINU(2) LYRA_ROR_MINU(1)
#undef LYRA_ROR_MINU
    const float m = __builtin_bit_cast(float, mbits);
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(sum == m);
    const int best = __builtin_ctz((unsigned)(holders >> (tid & 48)) | 0x10000u) & 15;   // (& 15: NaN distances only)
    {  // r <- r - (r + (q - r)), the graph's three separate fp32 ops, on this lane's four dimensions
      const float* c = cbs[win % 3] + u * 16 * ROW;
      const f32x4 qv = *reinterpret_cast<const f32x4*>(&c[best * ROW + j * 4]);
      const f32x4 t1 = qv - mine;
      const f32x4 t2 = mine + t1;
      mine = mine - t2;
    }
    if (j == 0 && live) {
      if (indices) indices[(size_t)frame * 46 + k] = best;
      if (packets) {
        if (k & 1) packets[(size_t)frame * nbytes + (k >> 1)] = (uint8_t)(cur | best);
        else cur = best << 4;
      }
    }
  };
#pragma unroll 1
  for (int k = 0; k < num_stages; k += 2) {
    stage(k, rowa, PREFETCH ? rowb : rowa);
    if (k + 1 < num_stages) stage(k + 1, PREFETCH ? rowb : rowa, rowa);
  }
  if (j == 0 && frame < B && packet_bytes) packet_bytes[frame] = live ? nbytes : 0;
  if (j == 0 && live) {
    if (packets && (num_stages & 1)) packets[(size_t)frame * nbytes + (num_stages >> 1)] = (uint8_t)cur;
    if (indices)
      for (int k = num_stages; k < 46; ++k) indices[(size_t)frame * 46 + k] = -1;
  }
}

#endif   // LYRA_PARKED (chain form)

// =============================================================================================
// RVQ encode: replaces quantizer.tflite `encode` (555 ops) + the bit-string assembly of ResidualVectorQuantizer::Quantize
// (lyra/residual_vector_quantizer.cc:77-110) + Packet<>::Pack (lyra/packet.h:91-122).
// CERTIFIED SCREENING on the matrix pipe, the graph's exact chain only where the screen cannot decide.
//
// The reference's argmin is over S_k = the sequentially rounded fp32 sum above.  In exact arithmetic
// T_k = |r - c_k|^2 = |c_k|^2 - 2 r.c_k + |r|^2, and the 16 x 16 dot products of a stage's 16 codewords with 16 frames are
// ONE dense [16 x 64] x [64 x 16] GEMM: 16 v_mfma_f32_16x16x4_f32 per stage and wavefront instead of 4 x 192 dependent
// vector instructions per 4 frames.  A_k = N_k - 2 P_k (N_k = |c_k|^2 from the host, P_k the MFMA dot product) ranks the
// codewords up to a rounding error that is bounded rigorously (u = 2^-24, C = max_k |c_k| of the stage, any Rb >= |r|^2):
//     |S_k - T_k|           <= 67 u T_k                         (66 roundings on non-negative terms)
//     |A_k - (T_k - |r|^2)| <= 2 u |c_k|^2 + 142 u |r||c_k|     (N rounded once, <= 70 roundings on the dot product, one on A)
//     => S_j > S_k strictly whenever A_j - A_k > 280 u (|r| + C)^2, and (|r| + C)^2 <= 2 (Rb + C^2).
// The scores are compared as KEYS: bits(A_k + Rb + M) -- positive, so they order as integers -- with the low four bits
// replaced by k (a perturbation below 2^-19 of the score: one integer minimum yields winner AND index).  With the margin
//     M = 2^-14 (Rb + C^2)  >=  (560 u + 2^-16)(Rb + C^2)
// every codeword whose key exceeds the smallest key by more than M cannot be the reference's ARG_MIN.  If exactly one
// codeword is left it IS the reference's index -- certified, no exact sum computed.  Otherwise (near-ties, exact ties,
// NaNs: count != 1) the frame takes the exact chain -- the graph's three fp32 operations per term in ascending order, first
// minimum -- for all 16 codewords.  (On speech about one frame-stage in a thousand does: lyra_hip_debug_read(5).)
// Rb is carried from stage to stage: |r'|^2 <= T_winner + 7 u (..)^2 <= A_min + M + Rb + (error) => Rb' = (key_min + 2 M)(1 + 2^-10).
// The residual update r <- r - (r + (q - r)) is the graph's, bit for bit, so the residual the next stage sees -- and
// every index -- is the reference's.
//
// One wavefront = 16 frames = one MFMA N tile.  Lane (n = lane & 15, q = lane >> 4) owns dims 16 q .. 16 q + 15 of frame n
// (B operand of MFMA kk: dim 16 q + kk) and, as A operand, the same dims of codeword n -- both are plain 64-byte reads of
// the natural layouts, no repacking.  D[codeword 4 q + e][frame n]: a lane holds four codewords' scores of ONE frame, so
// the argmin is three minima in registers and a two-step all-reduce over the four 16-lane rows; the winner is then known
// to exactly the four lanes that hold the frame's residual.  Its dims come back from the A-fragment registers of lane
// (winner, q) by ds_bpermute: the codebook never passes through LDS or memory a second time, and the certified path has
// no barrier.
// =============================================================================================
// All-reduce over the four 16-lane rows of a wavefront, register to register: v_permlane16_swap (odd rows of the first
// operand <-> even rows of the second) pairs rows 0|1 and 2|3, v_permlane32_swap (upper half <-> lower half) pairs the halves.
__device__ __forceinline__ unsigned rows_min_u32(unsigned v) {
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = min(a[0], a[1]);
  const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // [lo, lo], [hi, hi]
  return min(s[0], s[1]);
}
__device__ __forceinline__ unsigned rows_add_u32(unsigned v) {
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] + a[1];
  const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return s[0] + s[1];
}

__global__ __launch_bounds__(64) void rvq_encode_kernel(const float* __restrict__ cb, const float* __restrict__ cbn,
                                                        const float* __restrict__ feats, int B, int num_stages,
                                                        int32_t* __restrict__ indices, uint8_t* __restrict__ packets,
                                                        const int32_t* __restrict__ mask_ids,
                                                        int32_t* __restrict__ packet_bytes, unsigned* __restrict__ stats) {
  LYRA_STRESS(3);
  __shared__ __attribute__((aligned(16))) float rs[16 * 68];   // residuals of the tile, [frame][64 (+4 pad)]: exact path / prologue
  __shared__ int win[16];                                       // exact path: winners by frame
#ifdef LYRA_RVQ_PRIO   // experiment: the quantizer's lone wavefronts win the issue arbitration against the co-resident stage kernels
  __builtin_amdgcn_s_setprio(LYRA_RVQ_PRIO);
#endif
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4;
  const int x16 = (lane ^ 16) << 2;
  const int frame = blockIdx.x * 16 + n;
  const int f = min(frame, B - 1);
  const bool live = frame < B && !(mask_ids && mask_ids[f] < 0);
  const float LYRA_GLOBAL* cbg = as_global(cb) + (size_t)n * 64 + q * 16;   // this lane's 16 dims of codeword n, stage 0
  const float kUp = 1.0009765625f;   // 1 + 2^-10
  f32x4 r4[4], b4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) r4[i] = *reinterpret_cast<const f32x4*>(&feats[(size_t)f * 64 + q * 16 + i * 4]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b4[i] = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(cbg + i * 4);
  float Rb;   // >= |r|^2 of frame n (the same in its four lanes)
  {
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) part = __builtin_fmaf(r4[i][e], r4[i][e], part);
    part = part + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(x16, __builtin_bit_cast(int, part)));
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
    Rb = (__uint_as_float(s[0]) + __uint_as_float(s[1])) * kUp;
  }
  const int nbytes = (num_stages + 1) >> 1;
  int cur = 0;
  // Two fragment sets: stage k multiplies with set k & 1 while the winner of stage k - 1 is still read out of the other;
  // that one is then refilled with stage k + 1 (L2, in flight under the whole stage).
  f32x4 fa[4], fb[4], Na, Nb;
  float ca, cbb;
  int vz = 0;
  asm volatile("" : "+v"(vz));   // a VECTOR load for the stage scalar: a scalar load would share the LDS counter the bpermutes wait on
  auto fetch = [&](int k, f32x4 (&fr)[4], f32x4& Nn, float& c2s) {
    const int kc = min(k, 45);
    const float LYRA_GLOBAL* g = cbg + (size_t)kc * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) fr[i] = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(g + i * 4);
    Nn = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(cbn) + kc * 16 + q * 4);   // |c|^2 of codewords 4 q + e
    c2s = as_global(cbn)[46 * 16 + kc + vz];                                                // 2^-14 C^2, rounded up
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = b4[i];
  Na = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(cbn) + q * 4);
  ca = as_global(cbn)[46 * 16];
  fetch(1, fb, Nb, cbb);
  // the screen of one stage: P[e] = c_{4q+e} . r of frame n  ->  the stage's index (certified, or from the exact chain)
  auto screen = [&](int k, const f32x4& P, const f32x4& Nn, float c2s) -> int {
    const float M = __builtin_fmaf(Rb, 0x1p-14f, c2s);
    const float off = Rb + M;
    unsigned key[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = __builtin_fmaf(-2.f, P[e], Nn[e]) + off;   // > 0 (or NaN / huge: then nothing is certified)
      key[e] = (__float_as_uint(a) & ~15u) | (unsigned)(q * 4 + e);
    }
    const unsigned kmin = rows_min_u32(min(min(key[0], key[1]), min(key[2], key[3])));
    const float thr = __uint_as_float(kmin) + M;
    unsigned cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) cnt += __uint_as_float(key[e]) <= thr ? 1u : 0u;
    cnt = rows_add_u32(cnt);
    int best = (int)(kmin & 15u);
    // a score that is not an ordinary positive float (negative beyond the bound, NaN, Inf) certifies nothing
    const bool amb = cnt != 1u || !(__uint_as_float(kmin) < 0x1p126f) || (int)kmin < 0;
    const unsigned long long any_amb = __builtin_amdgcn_ballot_w64(amb);
    if (any_amb) {   // (wave-uniform) the exact chain for the frames the screen could not certify
      unsigned fm = (unsigned)any_amb & 0xffffu;   // a frame's four lanes agree
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&rs[n * 68 + q * 16 + i * 4]) = r4[i];
      __syncthreads();
      if (stats && lane == 0) { atomicAdd(&stats[0], (unsigned)__builtin_popcount(fm)); atomicAdd(&stats[1], 1u); }
      while (fm) {   // four flagged frames per pass: lane (slot = q, codeword = n)
        unsigned rest = fm;
        int fr = -1;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          const int bit = rest ? __builtin_ctz(rest) : -1;
          if (sl == q) fr = bit;
          if (rest) rest &= rest - 1;
        }
        fm = rest;
        const float* rr = &rs[max(fr, 0) * 68];
        const float LYRA_GLOBAL* cc = as_global(cb) + ((size_t)k * 16 + n) * 64;
        float sum = 0.f;
#pragma unroll 2
        for (int d4 = 0; d4 < 16; ++d4) {   // (rare path: kept small, its registers must not set the kernel's footprint)
          const f32x4 x = *reinterpret_cast<const f32x4*>(rr + d4 * 4);
          const f32x4 c = *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(cc + d4 * 4);
          const f32x4 df = x - c;
          const f32x4 sq = df * df;
          sum = sum + sq[0]; sum = sum + sq[1]; sum = sum + sq[2]; sum = sum + sq[3];
        }
        // ARG_MIN = first minimum (sums of squares: non-negative floats order like their bit patterns)
        unsigned mbits = __builtin_bit_cast(unsigned, sum);
#define LYRA_ROR_MINU(N) \
        asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(mbits) : "v"(mbits));
        LYRA_ROR_MINU(8) LYRA_ROR_MINU(4) LYRA_ROR_MINU(2) LYRA_ROR_MINU(1)
#undef LYRA_ROR_MINU
        const unsigned long long holders = __builtin_amdgcn_ballot_w64(sum == __builtin_bit_cast(float, mbits));
        const int xbest = __builtin_ctz((unsigned)(holders >> (lane & 48)) | 0x10000u) & 15;   // (& 15: NaN distances only)
        if (n == 0 && fr >= 0) win[fr] = xbest;
      }
      __syncthreads();
      if (amb) best = win[n];
      __syncthreads();   // win[] and rs[] are free again
    }
    Rb = __builtin_fmaf(2.f, M, __uint_as_float(kmin)) * kUp;
    if (q == 0 && live) {
      if (indices) indices[(size_t)frame * 46 + k] = best;
      if (packets) {
        if (k & 1) packets[(size_t)frame * nbytes + (k >> 1)] = (uint8_t)(cur | best);
        else cur = best << 4;
      }
    }
    return best;
  };
  // stage k >= 1: the residual update of stage k - 1 -- r <- r - (r + (q - r)), the graph's three separate fp32 ops, the
  // winner's dims out of lane (codeword best, q)'s fragments -- feeds this stage's MFMA chain dim by dim
  auto step = [&](int k, int best, f32x4 (&prev)[4], f32x4& Nprev, float& cprev, const f32x4 (&curf)[4], const f32x4& Ncur,
                  float ccur) -> int {
    const int src = (q * 16 + best) * 4;
    f32x4 qv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i32x4 bi = __builtin_bit_cast(i32x4, prev[i]);
      const i32x4 qi = {__builtin_amdgcn_ds_bpermute(src, bi[0]), __builtin_amdgcn_ds_bpermute(src, bi[1]),
                        __builtin_amdgcn_ds_bpermute(src, bi[2]), __builtin_amdgcn_ds_bpermute(src, bi[3])};
      qv[i] = __builtin_bit_cast(f32x4, qi);
    }
    __builtin_amdgcn_sched_barrier(0);   // all 16 in flight at once (the scheduler otherwise trickles them in pairs, each with its own wait)
    fetch(k + 1, prev, Nprev, cprev);   // (the bpermutes above have read `prev`)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 P[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // two chains: a dependent MFMA waits ~50 cycles, an independent one 32
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t1 = qv[i][e] - r4[i][e];
        const float t2 = r4[i][e] + t1;
        r4[i][e] = r4[i][e] - t2;
        P[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(curf[i][e], r4[i][e], P[e & 1], 0, 0, 0);
      }
    return screen(k, P[0] + P[1], Ncur, ccur);
  };
  int best;
  {
    f32x4 P[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      P[kk & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kk >> 2][kk & 3], r4[kk >> 2][kk & 3], P[kk & 1], 0, 0, 0);
    best = screen(0, P[0] + P[1], Na, ca);
  }
#pragma unroll 1
  for (int k = 1; k < num_stages; k += 2) {
    best = step(k, best, fa, Na, ca, fb, Nb, cbb);
    if (k + 1 < num_stages) best = step(k + 1, best, fb, Nb, cbb, fa, Na, ca);
  }
  if (q == 0 && frame < B && packet_bytes) packet_bytes[frame] = live ? nbytes : 0;
  if (q == 0 && live) {
    if (packets && (num_stages & 1)) packets[(size_t)frame * nbytes + (num_stages >> 1)] = (uint8_t)cur;
    if (indices)
      for (int k = num_stages; k < 46; ++k) indices[(size_t)frame * 46 + k] = -1;
  }
}

#ifdef LYRA_PARKED   // the all-exact chain kernels of rounds 2-3: bit-identical, kept for A/B in the variant build
// windows of two stages (26 KB of LDS), one register set (<= 128 VGPRs)
__global__ __launch_bounds__(256, 4) void rvq_encode_chain_kernel(const float* __restrict__ cb, const float* __restrict__ feats,
                                                            int B, int num_stages, int32_t* __restrict__ indices,
                                                            uint8_t* __restrict__ packets,
                                                            const int32_t* __restrict__ mask_ids,
                                                            int32_t* __restrict__ packet_bytes) {
  rvq_encode_body<2, false>(cb, feats, B, num_stages, indices, packets, mask_ids, packet_bytes);
}
// the 104 KB / 244-VGPR form
__global__ __launch_bounds__(256) void rvq_encode_wide_kernel(const float* __restrict__ cb, const float* __restrict__ feats,
                                                              int B, int num_stages, int32_t* __restrict__ indices,
                                                              uint8_t* __restrict__ packets,
                                                              const int32_t* __restrict__ mask_ids,
                                                              int32_t* __restrict__ packet_bytes) {
  rvq_encode_body<8, true>(cb, feats, B, num_stages, indices, packets, mask_ids, packet_bytes);
}
#endif

// RVQ decode: quantizer.tflite `decode` (233 ops) + the index extraction of DecodeToLossyFeatures
// (residual_vector_quantizer.cc:140-157).  ((v0 + v1) + v2) + ... strictly left to right, masked
// stages contribute v * 0.0f exactly as the graph does.
__global__ __launch_bounds__(256) void rvq_decode_kernel(const float* __restrict__ cb,
                                                          const int32_t* __restrict__ indices,
                                                          const uint8_t* __restrict__ packets, int num_stages, int B,
                                                          float* __restrict__ feats) {
  LYRA_STRESS(11);
  const int d = threadIdx.x & 63;
  const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (frame >= B) return;
  const int nbytes = (num_stages + 1) >> 1;
  float acc = 0.f;
#pragma unroll 2
  for (int k = 0; k < 46; ++k) {
    int id;
    if (packets) id = k < num_stages ? ((packets[(size_t)frame * nbytes + (k >> 1)] >> ((k & 1) ? 0 : 4)) & 15) : -1;
    else id = indices[(size_t)frame * 46 + k];
    float mask = id != -1 ? 1.f : 0.f;
    int i = id < 0 ? 0 : id;
    float v = cb[((size_t)k * 16 + i) * 64 + d] * mask;
    acc = k == 0 ? v : acc + v;
  }
  feats[(size_t)frame * 64 + d] = acc;
}

// =============================================================================================
// log-mel: LogMelSpectrogramExtractorImpl::Extract (lyra/log_mel_spectrogram_extractor_impl.cc:96-126)
// as instantiated by NoiseEstimator (16 kHz, hop 320, window 640, 160 bands).
// One workgroup per PAIR of stream-frames: the two real windows are packed into one complex sequence
// (re = frame A, im = frame B), transformed by ONE fp64 FFT-1024 in five radix-4 passes in LDS (digit-reversed
// input, host-built twiddles), and separated again by the conjugate symmetry of real spectra -- a quarter of the
// butterfly passes per frame of a per-frame radix-2 transform.  Then |X|, one thread per mel band accumulating its
// bins in ascending order (== the reference's scatter loop order per band), log / floor.
// The spectrum differs from the oracle's radix-2 one in the last bits of the doubles; after the cast to float the
// two agree except when a band sum lies within ~1e-16 of a float rounding boundary.
// `state` / `stride` / `prev_off`: where the previous hop of each stream lives -- the plugin-level extractor's own
// region (R_MEL) or the slot of one of the two NoiseEstimators (R_NOISE_E / R_NOISE_D own their extractor).
// =============================================================================================
// NoiseEstimator::ReceiveSamples' decision + recurrence for one stream per wavefront (defined below, next to its notes)
// Everything noise_update_wave reads from a stream's slot, requested in ONE batch (noise_prefetch) ahead of the work
// whose result it is compared with: lane l of the stream's wavefront owns bins l, l + 64, l + 128.
struct NoisePre { float est[3], bound[3], sm[3], sq[3], tm[3]; int initialised, hops; };
__device__ __forceinline__ NoisePre noise_prefetch(int id, const uint8_t* state, bool mine);
template <int NW>
__device__ __forceinline__ void noise_update_wave(const NoiseP& P, int w, bool on, int id, int out_index, uint8_t* state,
                                                  const float* mel, const NoisePre& pre, float* sh, float* avg,
                                                  int32_t* is_noise_out, int32_t* masked_ids);

// -DLYRA_MEL_ABL=bits: TIMING-ONLY ablations of the estimator kernel (results are wrong): 1 no log, 2 no exp, 4 no sqrt,
// 8 no decision / recurrence tail, 16 the kernel returns at once, 32 no radix-4 passes in LDS, 64 one bin per band
// (round 6, profiles/r06_ab_mel_ablation.txt)
#ifndef LYRA_MEL_ABL
#define LYRA_MEL_ABL 0
#endif
size_t logmel_lds_bytes() { return (size_t)1024 * 2 * 8; }             // the FFT buffer; everything later aliases dead parts of it
size_t cng_lds_bytes() { return (size_t)(1024 * 2 + 160) * 8; }      // (+160: the comfort-noise kernel's mel vector)

__device__ __forceinline__ int digit_reverse4_1024(int n) {   // reverse the five base-4 digits of n
  unsigned r = __brev((unsigned)n) >> 22;
  return (int)(((r & 0x2AAu) >> 1) | ((r & 0x155u) << 1));
}

// noise_tail != 0: `state` is a NoiseEstimator region and the kernel goes on with NoiseEstimator::ReceiveSamples' second
// half (noise_update_wave; wavefront f handles frame f) on the mel vector while it is still in LDS -- one launch and no
// trip of the 160 bins through HBM (`mel` may then be null).
__global__ __launch_bounds__(256) void logmel_kernel(const MelP* __restrict__ Pp, const int16_t* __restrict__ pcm,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, int stride, int prev_off,
                                                      float* __restrict__ mel, int noise_tail, NoiseP NP,
                                                      int32_t* __restrict__ is_noise_out,
                                                      int32_t* __restrict__ masked_ids) {
  LYRA_STRESS(7);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const MelP& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  f64x2* z = reinterpret_cast<f64x2*>(dsm);   // Z[1024], (re, im) = (frame A, frame B) interleaved; later (|X_A[k]|, |X_B[k]|)
  // LDS reuse once the two spectra are separated (z[k], k > 512, is then dead): the mel weights of bins 0..512, the hop's
  // mel vectors [2][160] floats and noise_update_wave's [2][2][160] + [2][2] floats -- 16 KB per workgroup in all
  double* wl = dsm + 1026;
  float* mel_lds = reinterpret_cast<float*>(dsm + 1540);
  float* tail_sh = reinterpret_cast<float*>(dsm + 1540 + 160);
  float* tail_avg = reinterpret_cast<float*>(dsm + 1540 + 160 + 320);
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * 2, b1 = b0 + 1;
  const bool two = b1 < B;
  if constexpr ((LYRA_MEL_ABL & 16) != 0) return;
  LYRA_TSTAMP(110);
  int16_t* prev0 = reinterpret_cast<int16_t*>(state + (size_t)ids[b0] * stride + prev_off);
  int16_t* prev1 = reinterpret_cast<int16_t*>(state + (size_t)ids[two ? b1 : b0] * stride + prev_off);
  const int16_t* cur0 = pcm + (size_t)b0 * 320;
  const int16_t* cur1 = pcm + (size_t)(two ? b1 : b0) * 320;
  // window = [previous hop | this hop] x periodic Hann, zero-padded to 1024.  The first radix-4 pass of a decimation-in-
  // time transform combines x[n], x[n+256], x[n+512], x[n+768]: thread n reads its own three window samples of both
  // frames (x[n+768] = 0; x[n+512] = 0 for n >= 128) straight from memory, does that butterfly in registers and writes
  // the four results where the second pass expects them (4 * digit-reverse(n) + q) -- no staging pass, no scatter of
  // single samples.  The samples of this hop it holds are exactly the stream's next history: thread n >= 64 holds
  // sample n-64 of the hop, thread n < 128 sample n+192.
  const int n = tid;
  const bool has1 = n >= 64, has2 = n < 128;
  const int16_t a0 = prev0[n], a1 = prev1[n];
  const int16_t b0s = has1 ? cur0[n - 64] : prev0[n + 256], b1s = has1 ? cur1[n - 64] : prev1[n + 256];
  const int16_t c0s = has2 ? cur0[n + 192] : (int16_t)0, c1s = has2 ? cur1[n + 192] : (int16_t)0;
  const double h0 = P.hann[n], h1 = P.hann[n + 256], h2 = has2 ? P.hann[n + 512] : 0.0;
  // twiddles of the second pass (L = 4), requested before the first butterfly; mel weights and band edges for the epilogue
  double w1r, w1i, w2r, w2i, w3r, w3i;
  {
    const int t1 = (tid & 3) * 64;
    w1r = P.tw4_re[t1]; w1i = P.tw4_im[t1]; w2r = P.tw4_re[2 * t1]; w2i = P.tw4_im[2 * t1];
    w3r = P.tw4_re[3 * t1]; w3i = P.tw4_im[3 * t1];
  }
  const double wsel0 = P.w[tid], wsel1 = P.w[tid + 256], wsel2 = tid == 0 ? P.w[512] : 0.0;
  // band sums: 320 (frame, band) items on 256 threads -- thread t < 160 takes (frame 0, band t), thread t >= 160 takes
  // (frame 1, band t - 96) i.e. the 96 widest bands, and threads t < 64 then also take (frame 1, band t), the narrow ones:
  // the longest chain is ONE wide band (<= 18 bins)
  const int my_band = tid < 160 ? tid : tid - 96;
  const int be0 = P.band[my_band], be1 = P.band[my_band + 1], be2 = P.band[my_band + 2];
  {
    const double ar = (double)a0 * h0, ai = two ? (double)a1 * h0 : 0.0;
    const double br = (double)b0s * h1, bi = two ? (double)b1s * h1 : 0.0;
    const double cr = (double)c0s * h2, ci = two ? (double)c1s * h2 : 0.0;
    const double s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;   // d = 0: b + d = b - d = b
    unsigned r = __brev((unsigned)n) >> 24;                                     // reverse the four base-4 digits of n
    r = ((r & 0xAAu) >> 1) | ((r & 0x55u) << 1);
    f64x2* o = z + 4 * r;
    o[0] = (f64x2){s0r + br, s0i + bi};
    o[1] = (f64x2){s1r + bi, s1i - br};     // (a - c) - i b
    o[2] = (f64x2){s0r - br, s0i - bi};
    o[3] = (f64x2){s1r - bi, s1i + br};     // (a - c) + i b
  }
  __syncthreads();
  // every read of the old history is done: the hop becomes the history
  if (has1) { prev0[n - 64] = b0s; if (two) prev1[n - 64] = b1s; }
  if (has2) { prev0[n + 192] = c0s; if (two) prev1[n + 192] = c1s; }
  LYRA_TSTAMP(111);
  // four more radix-4 passes; pass s combines four L-point transforms (L = 4^s) into one 4L-point one:
  //   y_q = sum_r (-i)^(r q) W_4L^(r k) F_r[k].  The twiddles of pass s+1 (L2-resident table) are requested before
  // the butterflies of pass s, so their latency hides behind the LDS round trip and the barrier.
#pragma unroll
  for (int s = 1; s < ((LYRA_MEL_ABL & 32) ? 1 : 5); ++s) {
    const int L = 1 << (2 * s);
    const int k = tid & (L - 1), g = tid >> (2 * s);
    f64x2* p = z + g * 4 * L + k;
    double n1r = 1.0, n1i = 0.0, n2r = 1.0, n2i = 0.0, n3r = 1.0, n3i = 0.0;
    if (s < 4) {
      const int Ln = 4 * L, kn = tid & (Ln - 1);
      const int t1 = kn * (256 >> (2 * (s + 1)));     // W_4L^k = W_1024^(k * 1024 / 4L)
      n1r = P.tw4_re[t1]; n1i = P.tw4_im[t1]; n2r = P.tw4_re[2 * t1]; n2i = P.tw4_im[2 * t1];
      n3r = P.tw4_re[3 * t1]; n3i = P.tw4_im[3 * t1];
    }
    const f64x2 xa = p[0], x1 = p[L], x2 = p[2 * L], x3 = p[3 * L];
    const double br = __builtin_fma(x1.x, w1r, -(x1.y * w1i)), bi = __builtin_fma(x1.x, w1i, x1.y * w1r);
    const double cr = __builtin_fma(x2.x, w2r, -(x2.y * w2i)), ci = __builtin_fma(x2.x, w2i, x2.y * w2r);
    const double dr = __builtin_fma(x3.x, w3r, -(x3.y * w3i)), di = __builtin_fma(x3.x, w3i, x3.y * w3r);
    const double s0r = xa.x + cr, s0i = xa.y + ci, s1r = xa.x - cr, s1i = xa.y - ci;
    const double s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
    p[0] = (f64x2){s0r + s2r, s0i + s2i};
    p[L] = (f64x2){s1r + s3i, s1i - s3r};         // (a - c) - i (b - d)
    p[2 * L] = (f64x2){s0r - s2r, s0i - s2i};
    p[3 * L] = (f64x2){s1r - s3i, s1i + s3r};     // (a - c) + i (b - d)
    w1r = n1r; w1i = n1i; w2r = n2r; w2i = n2i; w3r = n3r; w3i = n3i;
    __syncthreads();
  }
  LYRA_TSTAMP(112);
  // Z = FFT(a + i b):  A[k] = (Z[k] + conj(Z[N-k])) / 2,  B[k] = (Z[k] - conj(Z[N-k])) / (2 i).  In place: the item
  // for bin k <= 512 reads Z[k] and Z[N - k] and writes index k only; index k < 512 is read by no other item and
  // indices > 512 are never written, so no staging buffer (and no barrier before the writes) is needed.
  for (int k = tid; k <= 512; k += 256) {
    const int nk = (1024 - k) & 1023;
    const f64x2 zz = z[k], yy = z[nk];
    const double Ar = 0.5 * (zz.x + yy.x), Ai = 0.5 * (zz.y - yy.y);
    const double Br = 0.5 * (zz.y + yy.y), Bi = 0.5 * (yy.x - zz.x);
    if constexpr ((LYRA_MEL_ABL & 4) != 0) z[k] = (f64x2){Ar * Ar + Ai * Ai, Br * Br + Bi * Bi};
    else z[k] = (f64x2){__builtin_sqrt(Ar * Ar + Ai * Ai), __builtin_sqrt(Br * Br + Bi * Bi)};
  }
  __syncthreads();
  LYRA_TSTAMP(113);
  // the mel weights go to the now dead upper half of the buffer: the band loops below would otherwise wait for one L2
  // round trip per bin
  wl[tid] = wsel0;
  wl[tid + 256] = wsel1;
  if (tid == 0) wl[512] = wsel2;
  __syncthreads();
  LYRA_TSTAMP(114);
  // the noise tail's view of the stream's slot is requested here, one batch, and arrives under the band sums
  const int tw = tid >> 6;
  const int tail_id = ids[(tw == 1 && two) ? b1 : b0];
  NoisePre npre = {};
  if (noise_tail) npre = noise_prefetch(tail_id, state, tw < 2);
  // bins whose lower band is b-1 contribute (v - v*w) to band b, bins whose lower band is b contribute v*w; ascending
  // bin order (== the reference's scatter loop order per band); the next bin's operands are read one trip ahead
  auto band_item = [&](int f, int band) {
    const double* mag = dsm + f;          // |X_f[i]| = mag[2 * i]
    double acc = 0.0;
    double v = mag[2 * be0], wv = wl[be0];
    for (int i = be0; i < ((LYRA_MEL_ABL & 64) ? be0 + 1 : be2); ++i) {
      const double vn = mag[2 * i + 2], wn = wl[i + 1];
      const double w = v * wv;
      acc += i < be1 ? v - w : w;
      v = vn; wv = wn;
    }
    float x = (float)acc;
    x = x > 500.f ? x : 500.f;
    // log evaluated in double and rounded once: identical on host and device (oracle/lyra_oracle.c log_f)
    const float lm = (LYRA_MEL_ABL & 1) ? x * 1e-9f : (float)log((double)x) / 10.f;
    if (mel) mel[(size_t)(b0 + f) * 160 + band] = lm;
    if (noise_tail) mel_lds[f * 160 + band] = lm;
  };
  if (tid < 160) band_item(0, tid);
  else if (two) band_item(1, tid - 96);
  if (tid < 64 && two) band_item(1, tid);
  LYRA_TSTAMP(115);
  if (noise_tail && !(LYRA_MEL_ABL & 8)) {   // (uniform)
    __syncthreads();
    LYRA_TSTAMP2(120);
    const int w = tw;
    const bool on = w == 0 || (w == 1 && two);
    noise_update_wave<2>(NP, w, on, tail_id, b0 + (w & 1), state, mel_lds + (w & 1) * 160, npre, tail_sh, tail_avg,
                         is_noise_out, masked_ids);
  }
  LYRA_TSTAMP(119);
}

// =============================================================================================
// NoiseEstimator::ReceiveSamples after the log-mel (lyra/noise_estimator.cc:161-173): ComputeIsNoise, then
// DecayBounds or UpdateNoiseEstimate (SmoothingFactor, UpdateMinAndTemp, ComputeBounds), for one hop of B streams.
// One wavefront per stream, lane l owns bins l, l + 64, l + 128 (< 160).  Same float operations in the same order
// as the reference; the two Average() sums are sequential (std::accumulate from 0.f) and run on one lane out of
// LDS.  std::exp(float) is evaluated as float(exp(double)): a <= 1 ULP double result rounds to the correctly rounded
// float, which is what the host libm returns.  masked_ids (optional): ids[i], or -1 where the hop is noise -- the
// stream list the DTX-enabled encoder runs on (lyra_encoder.cc:131-141).
// =============================================================================================
__device__ __forceinline__ float expf_via_double(float x) {
  if constexpr ((LYRA_MEL_ABL & 2) != 0) return x * 0.5f;
  return (float)exp((double)x);
}

// One wavefront = one stream (slot w of the workgroup, NW slots); every thread of the workgroup calls this (two
// workgroup barriers inside).  `mel`: the hop's 160 log-mel bins (LDS or global); `on`: the slot holds a real stream.
__device__ __forceinline__ NoisePre noise_prefetch(int id, const uint8_t* state, bool mine) {
  const int lane = threadIdx.x & 63;
  const uint8_t* base = state + (size_t)id * st::NOISE_BYTES;
  const int* hdr = reinterpret_cast<const int*>(base);
  NoisePre p;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int bin = lane + 64 * i;
    const bool ld = mine && bin < 160;
    p.est[i] = ld ? reinterpret_cast<const float*>(base + st::N_EST)[bin] : 0.f;
    p.bound[i] = ld ? reinterpret_cast<const float*>(base + st::N_BOUND)[bin] : 0.f;
    p.sm[i] = ld ? reinterpret_cast<const float*>(base + st::N_SMOOTH)[bin] : 0.f;
    p.sq[i] = ld ? reinterpret_cast<const float*>(base + st::N_SQ)[bin] : 0.f;
    p.tm[i] = ld ? reinterpret_cast<const float*>(base + st::N_TMPMIN)[bin] : 0.f;
  }
  p.initialised = hdr[st::N_INIT / 4];
  p.hops = hdr[st::N_HOPS / 4];
  return p;
}

template <int NW>
__device__ __forceinline__ void noise_update_wave(const NoiseP& P, int w, bool on, int id, int out_index, uint8_t* state,
                                                  const float* mel, const NoisePre& pre, float* sh, float* avg,
                                                  int32_t* is_noise_out, int32_t* masked_ids) {
  // sh: [NW][2][160] floats, avg: [NW][2] floats of LDS scratch owned by the caller
  const int lane = threadIdx.x & 63;
  const bool mine = w < NW;          // waves beyond the NW slots only take part in the barriers
  const int ws = mine ? w : 0;
  uint8_t* base = state + (size_t)id * st::NOISE_BYTES;
  int* hdr = reinterpret_cast<int*>(base);
  float* f_smooth = reinterpret_cast<float*>(base + st::N_SMOOTH);
  float* f_sq = reinterpret_cast<float*>(base + st::N_SQ);
  float* f_tmp = reinterpret_cast<float*>(base + st::N_TMPMIN);
  float* f_est = reinterpret_cast<float*>(base + st::N_EST);
  float* f_bound = reinterpret_cast<float*>(base + st::N_BOUND);
  float cur[3], est[3], bound[3];
  bool differs = false;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int bin = lane + 64 * i;
    cur[i] = est[i] = bound[i] = 0.f;
    if (bin < 160 && mine) {
      cur[i] = mel[bin];
      est[i] = pre.est[i];
      bound[i] = pre.bound[i];
      differs = differs || (__builtin_fabsf(cur[i] - est[i]) > bound[i]);
    }
  }
  const bool is_noise = __builtin_amdgcn_ballot_w64(differs) == 0ull;   // ComputeIsNoise (wave-uniform)
  const int initialised = pre.initialised;
  const int hops = pre.hops;
  float sm[3], sq[3], tm[3];
  if (!is_noise && mine) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      sm[i] = sq[i] = tm[i] = 0.f;
      if (bin < 160) {
        if (initialised) { sm[i] = pre.sm[i]; sq[i] = pre.sq[i]; tm[i] = pre.tm[i]; }
        else { sm[i] = cur[i]; sq[i] = cur[i] * cur[i]; tm[i] = cur[i]; }   // first update (noise_estimator.cc:180-186)
        sh[(ws * 2 + 0) * 160 + bin] = sm[i];
        sh[(ws * 2 + 1) * 160 + bin] = cur[i];
      }
    }
  }
  if (NW == 2) LYRA_TSTAMP2(116);
  __syncthreads();
  // Average(): sequential float sum from 0.f, then / 160 -- one lane per sum.  In a workgroup with a wavefront to spare
  // (256 threads, NW < 4) that wavefront does all 2 * NW sums at once (of rows a noise hop left unwritten too: unused),
  // off the path of the wavefronts that carry the streams.
  constexpr bool kSpare = NW < 4;
  // the per-bin factor of SmoothingFactor() does not need the averages: evaluated here, beside the summing lanes
  float ebin[3] = {0.f, 0.f, 0.f};
  if (!is_noise && mine) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float de = (sm[i] - est[i]) / 0.3f;
      ebin[i] = expf_via_double(-(de * de));
    }
  }
  if (kSpare ? (w == NW && lane < 2 * NW) : (!is_noise && mine && lane < 2)) {
    const int row = kSpare ? lane : ws * 2 + lane;
    float a = 0.f;
    for (int i = 0; i < 160; ++i) a = a + sh[row * 160 + i];
    avg[row] = a / 160.f;
  }
  __syncthreads();
  if (NW == 2) LYRA_TSTAMP2(117);
  if (!on || !mine) return;
  if (is_noise) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      if (bin < 160) f_bound[bin] = bound[i] * P.bound_decay;   // DecayBounds
    }
  } else {
    const float kPowDiff = 0.3f;
    const float dd = (avg[ws * 2 + 0] - avg[ws * 2 + 1]) / kPowDiff;
    const float correction = expf_via_double(-(dd * dd));
    const double logn = 5.075173815233827;   // std::log(160) in double (noise_bound_.size())
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      if (bin < 160) {
        const float sf = P.max_smoothing * correction * ebin[i];
        const float c2 = cur[i] * cur[i];
        const float nsm = sf * sm[i] + (1.f - sf) * cur[i];   // (-ffp-contract=off: every product rounded)
        const float nsq = sf * sq[i] + (1.f - sf) * c2;
        float nest, ntm;
        if (hops == 0) { nest = __builtin_fminf(tm[i], nsm); ntm = nsm; }                      // UpdateMinAndTemp
        else { nest = __builtin_fminf(est[i], nsm); ntm = __builtin_fminf(tm[i], nsm); }
        float var = nsq - nsm * nsm;
        var = var > 0.f ? var : 0.f;
        f_smooth[bin] = nsm; f_sq[bin] = nsq; f_tmp[bin] = ntm; f_est[bin] = nest;
        f_bound[bin] = (float)((double)0.9f * __builtin_sqrt((double)var * logn));            // ComputeBounds
      }
    }
  }
  if (NW == 2) LYRA_TSTAMP2(118);
  if (lane == 0) {
    if (!is_noise) {
      hdr[st::N_INIT / 4] = 1;
      hdr[st::N_HOPS / 4] = (hops + 1) % P.hops_per_update;
    }
    hdr[st::N_IS_NOISE / 4] = is_noise ? 1 : 0;
    if (is_noise_out) is_noise_out[out_index] = is_noise ? 1 : 0;
    if (masked_ids) masked_ids[out_index] = is_noise ? -1 : id;
  }
}

// Stand-alone form (mel computed elsewhere): four streams per workgroup.
__global__ __launch_bounds__(256) void noise_update_kernel(NoiseP P, const int32_t* __restrict__ ids, int B,
                                                            uint8_t* __restrict__ state, const float* __restrict__ mel,
                                                            int32_t* __restrict__ is_noise_out,
                                                            int32_t* __restrict__ masked_ids) {
  LYRA_STRESS(7);
  const int w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  const bool on = b < B;
  const int bb = on ? b : B - 1;
  const int id = ids[bb];
  __shared__ float sh[4 * 2 * 160];
  __shared__ float avg[4 * 2];
  const NoisePre pre = noise_prefetch(id, state, true);
  noise_update_wave<4>(P, w, on, id, b, state, mel + (size_t)bb * 160, pre, sh, avg, is_noise_out, masked_ids);
}

// noise_estimate() / noise_bound() of B streams -> dense [B][160] (NoiseEstimator::noise_estimate, :229-231)
__global__ __launch_bounds__(256) void noise_read_kernel(const int32_t* __restrict__ ids, int B,
                                                          const uint8_t* __restrict__ state, int field_off,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 160) return;
  const int b = i / 160, bin = i - b * 160;
  out[i] = reinterpret_cast<const float*>(state + (size_t)ids[b] * st::NOISE_BYTES + field_off)[bin];
}

// =============================================================================================
// Resampler::Resample (lyra/resampler.cc:57-62): audio_dsp::QResampler<float> restated as a polyphase FIR sampled from
// a Kaiser-windowed sinc (oracle/lyra_oracle.c lo_resampler_design builds the same table; parity statement there).
// One wavefront per stream, four streams per workgroup: [34 history samples | n_in new samples] as floats in LDS, taps
// oldest first in float with separately rounded products -- bitwise the oracle's loop.  up-sampling: `up` outputs per
// input sample; down-sampling: one output per `down` inputs, phase carried in the stream's slot.
// =============================================================================================
int resample_streams_per_wg() { return 4; }
size_t resample_lds_bytes(int n_in) { return (size_t)4 * ((st::RS_TAPS - 1 + n_in + 3) & ~3) * 4; }

__global__ __launch_bounds__(256) void resample_kernel(ResampleP P, const int32_t* __restrict__ ids, int B,
                                                        uint8_t* __restrict__ state, const int16_t* __restrict__ in,
                                                        int n_in, int in_stride, int16_t* __restrict__ out, int n_out,
                                                        int out_stride) {
  LYRA_STRESS(8);
  extern __shared__ __attribute__((aligned(16))) float rsb_all[];   // [4][RS_TAPS - 1 + n_in, padded to 4]
  constexpr int H = st::RS_TAPS - 1;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x * 4 + w;
  const bool on = b < B;
  const int bb = on ? b : B - 1;
  float* rsb = rsb_all + w * ((H + n_in + 3) & ~3);
  const int16_t* src = in + (size_t)bb * in_stride;
  // the new samples do not wait for the stream id; 16-byte items when the rows allow it
  const bool vec = ((in_stride | n_in) & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  if (vec) {
    for (int c = lane; c * 8 < n_in; c += 64) {
      const i32x4 raw = *reinterpret_cast<const i32x4*>(src + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) rsb[H + c * 8 + e] = (float)(int16_t)((raw[e >> 1] >> ((e & 1) * 16)) & 0xffff);
    }
  } else {
    for (int i = lane; i < n_in; i += 64) rsb[H + i] = (float)src[i];
  }
  uint8_t* slot = state + (size_t)ids[bb] * st::RS_BYTES;
  float* hist = reinterpret_cast<float*>(slot + st::RS_HIST);
  const int in_pos = *reinterpret_cast<const int*>(slot + st::RS_IN_POS);
  if (lane < H) rsb[lane] = hist[lane];
  __syncthreads();
  if (!on) return;
  int16_t* dst = out + (size_t)b * out_stride;
  auto clip = [](float acc) { return (int16_t)(acc < -32768.f ? -32768.f : (acc > 32767.f ? 32767.f : acc)); };   // ClipToInt16 (dsp_utils.h:56-72)
  if (P.down == 1) {
    // interpolation: output k * up + ph is phase ph of the window at input k -- a lane takes input positions
    // k = lane, lane + 64, ...: one window of 35 samples in registers feeds all `up` phases, coefficients are scalars
    for (int k = lane; k < n_in; k += 64) {
      float win[st::RS_TAPS];
#pragma unroll
      for (int j = 0; j < st::RS_TAPS; ++j) win[j] = rsb[k + j];
#pragma unroll
      for (int ph = 0; ph < 3; ++ph) {
        if (ph < P.up) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < st::RS_TAPS; ++j) acc = acc + P.coef[ph][j] * win[j];
          dst[k * P.up + ph] = clip(acc);
        }
      }
    }
  } else {
    // decimation: the first input index (0-based in this call) that yields an output follows from the carried phase
    const int first = (P.down - in_pos % P.down) % P.down;
    for (int o = lane; o < n_out; o += 64) {
      const int k = first + o * P.down;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < st::RS_TAPS; ++j) acc = acc + P.coef[0][j] * rsb[k + j];
      dst[o] = clip(acc);
    }
  }
  // (same wavefront wrote and read this row: no barrier needed before the history leaves it)
  if (lane < H) hist[lane] = rsb[n_in + lane];
  // only the decimation phase is ever used: kept modulo 6 = lcm of the possible `down` factors (1, 2, 3), so the
  // counter never wraps out of phase however long the stream runs (the oracle keeps an unbounded counter)
  if (lane == 0) *reinterpret_cast<int*>(slot + st::RS_IN_POS) = (in_pos + n_in) % 6;
}

// =============================================================================================
// Device half of the batched LyraDecoder twin (host/lyra_batch_codec.cc).  The host runs the reference's per-stream
// state machine on integers only; the conditioned hops of the generative model and of the comfort-noise generator live
// in two [max_streams][320] arrays indexed by stream id, and each round's slices are assembled here.
// =============================================================================================
// dst[ids[b]][0..320) = src[b][0..320): a dense batch result into the by-id hop buffer (8 bytes per thread)
__global__ __launch_bounds__(128) void twin_scatter_kernel(const int16_t* __restrict__ src, const int32_t* __restrict__ ids,
                                                            int B, int16_t* __restrict__ dst) {
  LYRA_STRESS(10);
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < 80) reinterpret_cast<uint2*>(dst + (size_t)ids[b] * 320)[t] = reinterpret_cast<const uint2*>(src + (size_t)b * 320)[t];
}
// One workgroup per slice.  RunModel slices + MaybeOverlapAndInsert (lyra_decoder.cc:342-373): where both hops
// contribute, sample i is (int16)(gan * w + cng * (1.f - w)) with w = fade_w[fade + i * fade_dir] -- the table holds
// (1.f + std::cos(fade * M_PI / 640)) / 2.f evaluated on the host in double exactly as the reference writes it, so the
// mix is bit-identical to the reference's on that host; float products and sum are separate roundings
// (-ffp-contract=off), the conversion truncates as the implicit float -> int16_t of push_back does.
// noise_row >= 0: this slice completes a RECEIVED hop -- its 320 samples are what noise_estimator_->ReceiveSamples has
// accumulated (:304-311); they are copied to row noise_row of the dense buffer the estimator kernel then reads.
__global__ __launch_bounds__(256) void twin_assemble_kernel(const TwinSlice* __restrict__ slices, int B,
                                                             const int16_t* __restrict__ gan,
                                                             const int16_t* __restrict__ cng,
                                                             const float* __restrict__ fade_w, int16_t* __restrict__ out,
                                                             int out_stride, int16_t* __restrict__ noise_dense) {
  LYRA_STRESS(10);
  const TwinSlice s = slices[blockIdx.x];
  const int n = s.gen_n > s.cng_n ? s.gen_n : s.cng_n;
  const int16_t* g = gan + (size_t)s.id * 320 + s.gan_off;
  const int16_t* c = cng + (size_t)s.id * 320 + s.cng_off;
  int16_t* o = out + (size_t)s.id * out_stride + s.out_off;
  for (int i = threadIdx.x; i < n; i += 256) {
    int16_t v;
    if (s.cng_n == 0) v = g[i];
    else if (s.gen_n == 0) v = c[i];
    else {
      const float w = fade_w[s.fade + i * s.fade_dir - TWIN_FADE_LO];
      const float a = (float)g[i] * w;
      const float b = (float)c[i] * (1.f - w);
      v = (int16_t)(int)(a + b);
    }
    o[i] = v;
  }
  if (s.noise_row >= 0)
    for (int i = threadIdx.x; i < 80; i += 256)
      reinterpret_cast<uint2*>(noise_dense + (size_t)s.noise_row * 320)[i] =
          reinterpret_cast<const uint2*>(gan + (size_t)s.id * 320)[i];
}

// =============================================================================================
// ComfortNoiseGenerator::AddFeatures + GenerateSamples(320) (lyra/comfort_noise_generator.cc:74-119): log-mel -> mel ->
// estimated FFT magnitudes -> random phase -> inverse STFT (FFT 1024, step 320) -> ClipToInt16.  Restated construction,
// counter-based phases, parity statement in oracle/lyra_oracle.c (lo_cng_generate) -- this kernel follows that function
// operation by operation in fp64 (same radix-2 butterflies and host-built twiddles as the log-mel kernel).
// features == nullptr: the stream's decoder-side noise estimate is used (lyra_decoder.cc:328-340).
// =============================================================================================
__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void cng_kernel(const MelP* __restrict__ Pp, unsigned long long seed,
                                                   const int32_t* __restrict__ ids, int B,
                                                   uint8_t* __restrict__ state, const uint8_t* __restrict__ noise_state,
                                                   const float* __restrict__ features, int16_t* __restrict__ pcm) {
  LYRA_STRESS(9);
  const MelP& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* re = dsm;
  double* im = dsm + 1024;
  double* mel = dsm + 2048;   // [160], then unused
  const int tid = threadIdx.x, b = blockIdx.x;
  const int id = ids[b];
  uint8_t* slot = state + (size_t)id * st::CNG_BYTES;
  const unsigned long long hop = *reinterpret_cast<const unsigned long long*>(slot + st::C_HOP);
  double* ola = reinterpret_cast<double*>(slot + st::C_OLA);
  const float* feat = features ? features + (size_t)b * 160
                               : reinterpret_cast<const float*>(noise_state + (size_t)id * st::NOISE_BYTES + st::N_EST);
  if (tid < 160) mel[tid] = (double)(float)exp((double)(feat[tid] * 10.f));   // std::exp(float * kNorm), float
  for (int i = tid; i < 1024; i += 256) { re[i] = 0.0; im[i] = 0.0; }
  __syncthreads();
  const double PI = 3.14159265358979323846;
  const double gain = __builtin_sqrt(1024.0 * 320.0 / (384.0 * 240.0));
  const unsigned long long sd = seed ^ (unsigned long long)(unsigned)id;
  for (int i = P.start + tid; i <= P.end; i += 256) {
    // band[v + 1] = first bin whose lower band is >= v: find this bin's lower band ch (-1 .. 159)
    int lo = 0, hi = 161;   // band index domain v + 1
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (P.band[mid] <= i) lo = mid; else hi = mid; }
    const int ch = lo - 1;
    const double w = P.w[i];
    double v = 0.0;
    if (ch >= 0 && P.wsum[ch] > 0.0) v += w * mel[ch] / P.wsum[ch];
    if (ch + 1 < 160 && P.wsum[ch + 1] > 0.0) v += (1.0 - w) * mel[ch + 1] / P.wsum[ch + 1];
    const unsigned long long r = splitmix64_dev(sd ^ splitmix64_dev(hop * 1024ull + (unsigned long long)i));
    const double ang = (double)(r >> 11) * (1.0 / 9007199254740992.0) * 2.0 * PI;
    const double a = v * gain;
    const double xr = a * cos(ang), xi = a * sin(ang);
    // inverse DFT through the forward butterflies: conj in, conj out; inputs go to bit-reversed positions
    const int r0 = __brev((unsigned)i) >> 22;
    re[r0] = xr; im[r0] = (i == 0 || i == 512) ? 0.0 : -xi;
    if (i > 0 && i < 512) { const int r1 = __brev((unsigned)(1024 - i)) >> 22; re[r1] = xr; im[r1] = xi; }
  }
  __syncthreads();
#pragma unroll 1
  for (int p = 1; p <= 10; ++p) {
    const int len = 1 << p, half = len >> 1;
    for (int bf = tid; bf < 512; bf += 256) {
      int grp = bf >> (p - 1), k = bf & (half - 1);
      int i0 = grp * len + k, i1 = i0 + half;
      double wr = P.tw_re[half - 1 + k], wi = P.tw_im[half - 1 + k];
      double ur = re[i0], ui = im[i0];
      double xr = re[i1], xi = im[i1];
      double vr = xr * wr - xi * wi;
      double vi = xr * wi + xi * wr;
      re[i0] = ur + vr; im[i0] = ui + vi;
      re[i1] = ur - vr; im[i1] = ui - vi;
    }
    __syncthreads();
  }
  // window, overlap-add, emit the first 320 samples, shift the accumulator by one hop
  double acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = tid + 256 * q;
    const double x = re[n] / 1024.0;
    const double v = 0.5 - 0.5 * cos(2.0 * PI * n / 1024.0);
    acc[q] = ola[n] + x * v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = tid + 256 * q;
    if (n < 320) {
      double y = acc[q];
      y = y < -32768.0 ? -32768.0 : (y > 32767.0 ? 32767.0 : y);   // ClipToInt16<double>
      pcm[(size_t)b * 320 + n] = (int16_t)y;
    } else {
      re[n - 320] = acc[q];   // staging for the shifted write-back
    }
  }
  __syncthreads();
  for (int n = tid; n < 1024; n += 256) ola[n] = n < 1024 - 320 ? re[n] : 0.0;
  if (tid == 0) *reinterpret_cast<unsigned long long*>(slot + st::C_HOP) = hop + 1;
}

// =============================================================================================
// state reset: zeros everywhere (the graphs' CALL_ONCE init subgraph assigns zero constants), int8
// histories hold the zero point of their tensor (== quantize(0.0f)); NoiseEstimator: is_noise_ = true
// (lyra/noise_estimator.cc:131).  One workgroup per stream walks that stream's slot in every region.
// =============================================================================================
__global__ __launch_bounds__(256) void reset_kernel(const ResetP* __restrict__ Pp, const int32_t* __restrict__ ids, int n, int all,
                                                     StateMap sm) {
  const ResetP& P = *Pp;
  const int sidx = blockIdx.x;
  if (sidx >= n) return;
  const int id = all ? sidx : ids[sidx];
#pragma unroll 1
  for (int r = 0; r < st::R_COUNT; ++r) {
    const int bytes = sm.bytes[r];   // (st::REGION_BYTES is a host-side table: no runtime indexing of it here)
    uint8_t* base = sm.base[r] + (size_t)id * bytes;
    for (int o = threadIdx.x * 16; o < bytes; o += 256 * 16) {
      int v = 0;
      if (r == st::R_E2) {
        if (o >= st::E_R2_1 && o < st::E_R2_2) v = P.e_r2_1;
        else if (o >= st::E_R2_2 && o < st::E_D2) v = P.e_r2_2;
        else if (o >= st::E_D2 && o < st::E_BOTT) v = P.e_d2;
        else if (o >= st::E_BOTT && o < st::E_BOTT + 2 * 512) v = P.e_bott;
      } else if (r == st::R_D0) {
        if (o >= st::D_R0_0 && o < st::D_R0_1) v = P.d_r0_0;
        else if (o >= st::D_R0_1 && o < st::D_R0_2) v = P.d_r0_1;
        else if (o >= st::D_R0_2 && o < st::D_UP1) v = P.d_r0_2;
      }
      const int w = (v & 255) * 0x01010101;
      i32x4 q = (i32x4){w, w, w, w};
      if ((r == st::R_NOISE_E || r == st::R_NOISE_D) && o == 0) q[st::N_IS_NOISE / 4] = 1;
      *reinterpret_cast<i32x4*>(base + o) = q;
    }
  }
}

}  // namespace lyra
