// misc_kernels.hip -- residual vector quantizer, packet (un)packing, log-mel front end, state reset.
#include "kernels.h"

namespace lyra {

// =============================================================================================
// RVQ encode: replaces quantizer.tflite `encode` (555 ops) + the bit-string assembly of
// ResidualVectorQuantizer::Quantize (lyra/residual_vector_quantizer.cc:77-110) + Packet<>::Pack
// (lyra/packet.h:91-122).  16 lanes = the 16 codewords of a stage; each lane runs the 64-term
// squared-distance sum in the oracle's order (separate multiply and add, d ascending), then a
// 16-lane shuffle argmin with lowest-index tie break (ARG_MIN = first minimum).  The residual lives
// in LDS, shared by the 16 lanes of its frame (broadcast reads), and is updated once per dimension with the
// graph's three fp32 ops r - (r + (q - r)).  4 frames per wavefront, 16 per workgroup.  The 4 KB codebook of
// the current stage sits in LDS (rows padded to 68 floats: conflict-free ds_read_b128 across the 16 code lanes),
// double-buffered: the next stages' rows are fetched from L2 while this window computes.  63 VGPRs: the kernel
// is a 46-step dependent chain at one wavefront per SIMD, so what matters is how little it takes away from the
// decode-side kernels running next to it.
// =============================================================================================
__global__ __launch_bounds__(256) void rvq_encode_kernel(const float* __restrict__ cb,
                                                          const float* __restrict__ feats, int B, int num_stages,
                                                          int32_t* __restrict__ indices,
                                                          uint8_t* __restrict__ packets) {
  // Codebooks are staged through LDS in windows of W stages, double-buffered: window w+1 is fetched from L2
  // while window w computes (a whole window of compute hides the load latency; one barrier per window).
  constexpr int W = 4, WFLOATS = W * 16 * 68;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float cbs[2][WFLOATS];
  const int tid = threadIdx.x;
  const int j = tid & 15;
  const int frame = blockIdx.x * 16 + (tid >> 4);
  const int f = min(frame, B - 1);
  const int ldrow = tid >> 4, ldc4 = tid & 15;  // this thread's float4 of each stage's [16][64] codebook
  f32x4 nxt[W];
#pragma unroll
  for (int u = 0; u < W; ++u)
    nxt[u] = *reinterpret_cast<const f32x4*>(&cb[((size_t)min(u, 45) * 16 + ldrow) * 64 + ldc4 * 4]);
  // The residual of each frame lives in LDS (rs), not replicated in the registers of its 16 code lanes: the
  // distance pass reads it with broadcast ds_read_b128 next to the code rows, and the update r - (r + (q - r))
  // is done once per dimension (lane j owns dims 4j..4j+3) instead of sixteen times.  All 16 lanes of a frame sit
  // in one wavefront and LDS operations of a wavefront execute in order, so the write-back needs no barrier.
  __shared__ __attribute__((aligned(16))) float rs[16][64];
  float* rme = rs[tid >> 4];
  *reinterpret_cast<f32x4*>(&rme[j * 4]) = *reinterpret_cast<const f32x4*>(&feats[(size_t)f * 64 + j * 4]);
  const int nbytes = (num_stages + 1) >> 1;
  int cur = 0;
#pragma unroll 1
  for (int k = 0; k < num_stages; ++k) {
    const int u = k & (W - 1), win = k / W;
    if (u == 0) {
      // window `win` -> LDS (its loads were issued a window ago), then request window win+1
#pragma unroll
      for (int v = 0; v < W; ++v)
        *reinterpret_cast<f32x4*>(&cbs[win & 1][v * 16 * 68 + ldrow * 68 + ldc4 * 4]) = nxt[v];
#pragma unroll
      for (int v = 0; v < W; ++v)
        nxt[v] = *reinterpret_cast<const f32x4*>(&cb[((size_t)min((win + 1) * W + v, 45) * 16 + ldrow) * 64 + ldc4 * 4]);
      __syncthreads();  // window visible; (the previous barrier already guarantees nobody still reads this buffer:
                        //  it was last read two windows ago, and every thread passed the barrier in between)
    }
    const float* c = cbs[win & 1] + u * 16 * 68;
    asm volatile("" ::: "memory");   // rs is rewritten by the other lanes of the frame: never carry it in registers
    float sum = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 16; ++d4) {
      f32x4 cv = *reinterpret_cast<const f32x4*>(&c[j * 68 + d4 * 4]);
      f32x4 rv = *reinterpret_cast<const f32x4*>(&rme[d4 * 4]);
      f32x2 df0 = (f32x2){rv[0], rv[1]} - (f32x2){cv[0], cv[1]};
      f32x2 df1 = (f32x2){rv[2], rv[3]} - (f32x2){cv[2], cv[3]};
      f32x2 sq0 = df0 * df0, sq1 = df1 * df1;
      sum = sum + sq0[0];
      sum = sum + sq0[1];
      sum = sum + sq1[0];
      sum = sum + sq1[1];
    }
    int best = j;
    float bd = sum;
    // all-reduce over the 16 code lanes with DPP row rotations (row_ror:8/4/2/1): register-only, no LDS crossbar.
    // (distance, index) is totally ordered, so every lane ends with the same winner.
#define LYRA_ROR_STEP(N)                                                                                  \
  {                                                                                                       \
    float od = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bd),      \
                                                                       0x120 + (N), 0xf, 0xf, false));    \
    int oi = __builtin_amdgcn_update_dpp(0, best, 0x120 + (N), 0xf, 0xf, false);                          \
    if (od < bd || (od == bd && oi < best)) { bd = od; best = oi; }                                       \
  }
    LYRA_ROR_STEP(8) LYRA_ROR_STEP(4) LYRA_ROR_STEP(2) LYRA_ROR_STEP(1)
#undef LYRA_ROR_STEP
    {  // r <- r - (r + (q - r)), the graph's three separate fp32 ops, on this lane's four dimensions
      const f32x4 qv = *reinterpret_cast<const f32x4*>(&c[best * 68 + j * 4]);
      const f32x4 rv = *reinterpret_cast<const f32x4*>(&rme[j * 4]);
      const f32x4 t1 = qv - rv;
      const f32x4 t2 = rv + t1;
      *reinterpret_cast<f32x4*>(&rme[j * 4]) = rv - t2;
    }
    if (j == 0 && frame < B) {
      if (indices) indices[(size_t)frame * 46 + k] = best;
      if (packets) {
        if (k & 1) packets[(size_t)frame * nbytes + (k >> 1)] = (uint8_t)(cur | best);
        else cur = best << 4;
      }
    }
  }
  if (j == 0 && frame < B) {
    if (packets && (num_stages & 1)) packets[(size_t)frame * nbytes + (num_stages >> 1)] = (uint8_t)cur;
    if (indices)
      for (int k = num_stages; k < 46; ++k) indices[(size_t)frame * 46 + k] = -1;
  }
}

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
// One workgroup per stream-frame: fp64 radix-2 FFT-1024 in LDS (same butterfly order and twiddles as
// the oracle), |X|, then one thread per mel band accumulating its bins in ascending order (== the
// reference's scatter loop order per band), float log/floor.
// =============================================================================================
size_t logmel_lds_bytes() { return (size_t)(1024 * 2 + 520) * 8; }

__global__ __launch_bounds__(256) void logmel_kernel(const MelP* __restrict__ Pp, const int16_t* __restrict__ pcm,
                                                      const int32_t* __restrict__ ids, int B,
                                                      uint8_t* __restrict__ state, float* __restrict__ mel) {
  const MelP& P = *Pp;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* re = dsm;
  double* im = dsm + 1024;
  double* mag = dsm + 2048;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  int16_t* prev = reinterpret_cast<int16_t*>(state + (size_t)ids[b] * st::MEL_BYTES + st::M_PREV);
  for (int i = tid; i < 1024; i += 256) {
    double v = 0.0;
    if (i < 320) v = (double)prev[i] * P.hann[i];
    else if (i < 640) v = (double)pcm[(size_t)b * 320 + (i - 320)] * P.hann[i];
    int rv = __brev((unsigned)i) >> 22;
    re[rv] = v;
    im[rv] = 0.0;
  }
  __syncthreads();
  for (int i = tid; i < 320; i += 256) prev[i] = pcm[(size_t)b * 320 + i];
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
  for (int i = tid; i <= 512; i += 256) mag[i] = __builtin_sqrt(re[i] * re[i] + im[i] * im[i]);
  __syncthreads();
  if (tid < 160) {
    // bins whose lower band is tid-1 contribute (v - v*w); bins whose lower band is tid contribute v*w
    double acc = 0.0;
    for (int i = P.band[tid]; i < P.band[tid + 1]; ++i) { double v = mag[i]; double w = v * P.w[i]; acc += v - w; }
    for (int i = P.band[tid + 1]; i < P.band[tid + 2]; ++i) { double v = mag[i]; acc += v * P.w[i]; }
    float v = (float)acc;
    v = v > 500.f ? v : 500.f;
    mel[(size_t)b * 160 + tid] = __builtin_logf(v) / 10.f;
  }
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
