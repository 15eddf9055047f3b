// misc_kernels.hip -- residual vector quantizer, packet (un)packing, log-mel front end, state reset.
#include "kernels.h"

namespace lyra {

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
// RVQ encode: replaces quantizer.tflite `encode` (555 ops) + the bit-string assembly of
// ResidualVectorQuantizer::Quantize (lyra/residual_vector_quantizer.cc:77-110) + Packet<>::Pack
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

// The shipped form: windows of two stages (26 KB of LDS), one register set (<= 128 VGPRs): next to this kernel's one
// wavefront per SIMD three wavefronts of a stage kernel still fit, and the other side's stage kernel keeps most of its
// tiles resident while the quantizer runs (the 8-stage / two-register-set form is 104 KB and 244 VGPRs: faster alone,
// but it evicts the co-running kernel).
__global__ __launch_bounds__(256, 4) void rvq_encode_kernel(const float* __restrict__ cb, const float* __restrict__ feats,
                                                            int B, int num_stages, int32_t* __restrict__ indices,
                                                            uint8_t* __restrict__ packets,
                                                            const int32_t* __restrict__ mask_ids,
                                                            int32_t* __restrict__ packet_bytes) {
  rvq_encode_body<2, false>(cb, feats, B, num_stages, indices, packets, mask_ids, packet_bytes);
}
#ifdef LYRA_PARKED   // the 104 KB / 244-VGPR form: measured, not shipped
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
template <int NW>
__device__ __forceinline__ void noise_update_wave(const NoiseP& P, int w, bool on, int id, int out_index, uint8_t* state,
                                                  const float* mel, int32_t* is_noise_out, int32_t* masked_ids);

size_t logmel_lds_bytes() { return (size_t)(1024 * 2 + 160) * 8; }   // (+160: the comfort-noise kernel's mel vector)

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
  const MelP& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* re = dsm;
  double* im = dsm + 1024;
  double* mag0 = re;                  // |X_A[k]|, k = 0..512, written in place over Z (see below)
  double* mag1 = im;                  // |X_B[k]|
  float* mel_lds = reinterpret_cast<float*>(dsm + 2048);   // [2][160] floats in the 160 spare doubles (noise tail)
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * 2, b1 = b0 + 1;
  const bool two = b1 < B;
  int16_t* prev0 = reinterpret_cast<int16_t*>(state + (size_t)ids[b0] * stride + prev_off);
  int16_t* prev1 = reinterpret_cast<int16_t*>(state + (size_t)ids[two ? b1 : b0] * stride + prev_off);
  // window = [previous hop | this hop] x periodic Hann, zero-padded to 1024, in digit-reversed order.  Items of eight
  // samples (one 16-byte load each): 80 per frame; the previous hop is replaced in the same pass (each item rewrites
  // exactly the eight history samples it has just read, or none).
  for (int i = 640 + tid; i < 1024; i += 256) { const int r = digit_reverse4_1024(i); re[r] = 0.0; im[r] = 0.0; }
  if (tid < 160) {
    const int f = tid >= 80, c = tid - 80 * f;            // frame, chunk of 8 samples within the 640-sample window
    if (!f || two) {
      int16_t* prev = f ? prev1 : prev0;
      const int16_t* cur = pcm + (size_t)(f ? b1 : b0) * 320;
      const i32x4 raw = c < 40 ? *reinterpret_cast<const i32x4*>(prev + c * 8)
                               : *reinterpret_cast<const i32x4*>(cur + (c - 40) * 8);
      if (c < 40) *reinterpret_cast<i32x4*>(prev + c * 8) = *reinterpret_cast<const i32x4*>(cur + c * 8);
      double* dst = f ? im : re;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = c * 8 + e;
        const int16_t x = (int16_t)((raw[e >> 1] >> ((e & 1) * 16)) & 0xffff);
        dst[digit_reverse4_1024(i)] = (double)x * P.hann[i];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) im[digit_reverse4_1024(c * 8 + e)] = 0.0;
    }
  }
  __syncthreads();
  // five radix-4 decimation-in-time passes; pass s combines four L-point transforms (L = 4^s) into one 4L-point one:
  //   y_q = sum_r (-i)^(r q) W_4L^(r k) F_r[k].  The twiddles of pass s+1 (L2-resident table) are requested before
  // the butterflies of pass s, so their latency hides behind the LDS round trip and the barrier.
  double w1r = 1.0, w1i = 0.0, w2r = 1.0, w2i = 0.0, w3r = 1.0, w3i = 0.0;   // pass 0: L = 1, k = 0
  const double wsel0 = P.w[tid + 1], wsel1 = tid < 254 ? P.w[tid + 257] : 0.0;   // mel weights, parked for the epilogue
#pragma unroll 1
  for (int s = 0; s < 5; ++s) {
    const int L = 1 << (2 * s);
    const int k = tid & (L - 1), g = tid >> (2 * s);
    const int i0 = g * 4 * L + k, i1 = i0 + L, i2 = i1 + L, i3 = i2 + L;
    double n1r = 1.0, n1i = 0.0, n2r = 1.0, n2i = 0.0, n3r = 1.0, n3i = 0.0;
    if (s < 4) {
      const int Ln = 4 * L, kn = tid & (Ln - 1);
      const int t1 = kn * (256 >> (2 * (s + 1)));     // W_4L^k = W_1024^(k * 1024 / 4L)
      n1r = P.tw4_re[t1]; n1i = P.tw4_im[t1]; n2r = P.tw4_re[2 * t1]; n2i = P.tw4_im[2 * t1];
      n3r = P.tw4_re[3 * t1]; n3i = P.tw4_im[3 * t1];
    }
    const double ar = re[i0], ai = im[i0];
    const double xr1 = re[i1], xi1 = im[i1], xr2 = re[i2], xi2 = im[i2], xr3 = re[i3], xi3 = im[i3];
    const double br = xr1 * w1r - xi1 * w1i, bi = xr1 * w1i + xi1 * w1r;
    const double cr = xr2 * w2r - xi2 * w2i, ci = xr2 * w2i + xi2 * w2r;
    const double dr = xr3 * w3r - xi3 * w3i, di = xr3 * w3i + xi3 * w3r;
    const double s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
    const double s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
    re[i0] = s0r + s2r; im[i0] = s0i + s2i;
    re[i1] = s1r + s3i; im[i1] = s1i - s3r;     // (a - c) - i (b - d)
    re[i2] = s0r - s2r; im[i2] = s0i - s2i;
    re[i3] = s1r - s3i; im[i3] = s1i + s3r;     // (a - c) + i (b - d)
    w1r = n1r; w1i = n1i; w2r = n2r; w2i = n2i; w3r = n3r; w3i = n3i;
    __syncthreads();
  }
  // Z = FFT(a + i b):  A[k] = (Z[k] + conj(Z[N-k])) / 2,  B[k] = (Z[k] - conj(Z[N-k])) / (2 i).  In place: the item
  // for bin k <= 512 reads Z[k] and Z[N - k] and writes index k only; index k < 512 is read by no other item and
  // indices > 512 are never written, so no staging buffer (and no barrier before the writes) is needed.
  for (int k = tid; k <= 512; k += 256) {
    const int n = (1024 - k) & 1023;
    const double zr = re[k], zi = im[k], yr = re[n], yi = im[n];
    const double Ar = 0.5 * (zr + yr), Ai = 0.5 * (zi - yi);
    const double Br = 0.5 * (zi + yi), Bi = 0.5 * (yr - zr);
    mag0[k] = __builtin_sqrt(Ar * Ar + Ai * Ai);
    mag1[k] = __builtin_sqrt(Br * Br + Bi * Bi);
  }
  __syncthreads();
  // the mel weights of bins 1..510 go to the now unused upper half of `re` (index 513 + bin): the band loops below
  // would otherwise wait for one L2 round trip per bin
  double* wl = re + 513;
  wl[tid + 1] = wsel0;
  if (tid < 254) wl[tid + 257] = wsel1;
  __syncthreads();
  if (tid < 160) {
    const int e0 = P.band[tid], e1 = P.band[tid + 1], e2 = P.band[tid + 2];
#pragma unroll 1
    for (int f = 0; f < (two ? 2 : 1); ++f) {
      const double* mag = f ? mag1 : mag0;
      // bins whose lower band is tid-1 contribute (v - v*w); bins whose lower band is tid contribute v*w
      double acc = 0.0;
      for (int i = e0; i < e1; ++i) { double v = mag[i]; double w = v * wl[i]; acc += v - w; }
      for (int i = e1; i < e2; ++i) { double v = mag[i]; acc += v * wl[i]; }
      float v = (float)acc;
      v = v > 500.f ? v : 500.f;
      // log evaluated in double and rounded once: identical on host and device (oracle/lyra_oracle.c log_f)
      const float lm = (float)log((double)v) / 10.f;
      if (mel) mel[(size_t)(b0 + f) * 160 + tid] = lm;
      if (noise_tail) mel_lds[f * 160 + tid] = lm;
    }
  }
  if (noise_tail) {   // (uniform)
    __syncthreads();
    const int w = tid >> 6;
    const bool on = w == 0 || (w == 1 && two);
    noise_update_wave<2>(NP, w, on, ids[(w == 1 && two) ? b1 : b0], b0 + (w & 1), state, mel_lds + (w & 1) * 160,
                         is_noise_out, masked_ids);
  }
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
__device__ __forceinline__ float expf_via_double(float x) { return (float)exp((double)x); }

// One wavefront = one stream (slot w of the workgroup, NW slots); every thread of the workgroup calls this (two
// workgroup barriers inside).  `mel`: the hop's 160 log-mel bins (LDS or global); `on`: the slot holds a real stream.
template <int NW>
__device__ __forceinline__ void noise_update_wave(const NoiseP& P, int w, bool on, int id, int out_index, uint8_t* state,
                                                  const float* mel, int32_t* is_noise_out, int32_t* masked_ids) {
  __shared__ float sh[NW][2][160];
  __shared__ float avg[NW][2];
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
      est[i] = f_est[bin];
      bound[i] = f_bound[bin];
      differs = differs || (__builtin_fabsf(cur[i] - est[i]) > bound[i]);
    }
  }
  const bool is_noise = __builtin_amdgcn_ballot_w64(differs) == 0ull;   // ComputeIsNoise (wave-uniform)
  const int initialised = hdr[st::N_INIT / 4];
  const int hops = hdr[st::N_HOPS / 4];
  float sm[3], sq[3], tm[3];
  if (!is_noise && mine) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      sm[i] = sq[i] = tm[i] = 0.f;
      if (bin < 160) {
        if (initialised) { sm[i] = f_smooth[bin]; sq[i] = f_sq[bin]; tm[i] = f_tmp[bin]; }
        else { sm[i] = cur[i]; sq[i] = cur[i] * cur[i]; tm[i] = cur[i]; }   // first update (noise_estimator.cc:180-186)
        sh[ws][0][bin] = sm[i];
        sh[ws][1][bin] = cur[i];
      }
    }
  }
  __syncthreads();
  if (!is_noise && mine && lane < 2) {   // Average(): sequential float sum from 0.f, then / 160
    float a = 0.f;
    for (int i = 0; i < 160; ++i) a = a + sh[ws][lane][i];
    avg[ws][lane] = a / 160.f;
  }
  __syncthreads();
  if (!on || !mine) return;
  if (is_noise) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      if (bin < 160) f_bound[bin] = bound[i] * P.bound_decay;   // DecayBounds
    }
  } else {
    const float kPowDiff = 0.3f;
    const float dd = (avg[ws][0] - avg[ws][1]) / kPowDiff;
    const float correction = expf_via_double(-(dd * dd));
    const double logn = 5.075173815233827;   // std::log(160) in double (noise_bound_.size())
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int bin = lane + 64 * i;
      if (bin < 160) {
        const float de = (sm[i] - est[i]) / kPowDiff;
        const float sf = P.max_smoothing * correction * expf_via_double(-(de * de));
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
  const int w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  const bool on = b < B;
  const int bb = on ? b : B - 1;
  noise_update_wave<4>(P, w, on, ids[bb], b, state, mel + (size_t)bb * 160, is_noise_out, masked_ids);
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
// One workgroup per stream: [34 history samples | n_in new samples] as floats in LDS, one output per thread and
// iteration, taps oldest first in float -- bitwise the oracle's loop.  up-sampling: `up` outputs per input sample;
// down-sampling: one output per `down` inputs, phase carried in the stream's slot.
// =============================================================================================
__global__ __launch_bounds__(256) void resample_kernel(ResampleP P, const int32_t* __restrict__ ids, int B,
                                                        uint8_t* __restrict__ state, const int16_t* __restrict__ in,
                                                        int n_in, int in_stride, int16_t* __restrict__ out, int n_out,
                                                        int out_stride) {
  extern __shared__ __attribute__((aligned(16))) float rsb[];   // [RS_TAPS - 1 + n_in]
  constexpr int H = st::RS_TAPS - 1;
  const int b = blockIdx.x, tid = threadIdx.x;
  uint8_t* slot = state + (size_t)ids[b] * st::RS_BYTES;
  float* hist = reinterpret_cast<float*>(slot + st::RS_HIST);
  const int in_pos = *reinterpret_cast<const int*>(slot + st::RS_IN_POS);
  for (int i = tid; i < H + n_in; i += 256) rsb[i] = i < H ? hist[i] : (float)in[(size_t)b * in_stride + (i - H)];
  __syncthreads();
  // first input index (0-based in this call) that yields an output when decimating
  const int first = P.down == 1 ? 0 : ((P.down - in_pos % P.down) % P.down);
  for (int o = tid; o < n_out; o += 256) {
    int k, p;
    if (P.down == 1) { k = o / P.up; p = o - k * P.up; }
    else { k = first + o * P.down; p = 0; }
    float acc = 0.f;
#pragma unroll 5
    for (int j = 0; j < st::RS_TAPS; ++j) acc = acc + P.coef[p][j] * rsb[k + j];
    acc = acc < -32768.f ? -32768.f : (acc > 32767.f ? 32767.f : acc);   // ClipToInt16 (dsp_utils.h:56-72)
    out[(size_t)b * out_stride + o] = (int16_t)acc;
  }
  __syncthreads();
  for (int i = tid; i < H; i += 256) hist[i] = rsb[n_in + i];
  // only the decimation phase is ever used: kept modulo 6 = lcm of the possible `down` factors (1, 2, 3), so the
  // counter never wraps out of phase however long the stream runs (the oracle keeps an unbounded counter)
  if (tid == 0) *reinterpret_cast<int*>(slot + st::RS_IN_POS) = (in_pos + n_in) % 6;
}

// =============================================================================================
// Device half of the batched LyraDecoder twin (host/lyra_batch_codec.cc).  The host runs the reference's per-stream
// state machine on integers only; the conditioned hops of the generative model and of the comfort-noise generator live
// in two [max_streams][320] arrays indexed by stream id, and each round's slices are assembled here.
// =============================================================================================
// dst[ids[b]][0..320) = src[b][0..320): a dense batch result into the by-id hop buffer (8 bytes per thread)
__global__ __launch_bounds__(128) void twin_scatter_kernel(const int16_t* __restrict__ src, const int32_t* __restrict__ ids,
                                                            int B, int16_t* __restrict__ dst) {
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
