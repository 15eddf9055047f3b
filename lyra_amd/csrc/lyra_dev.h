// lyra_dev.h -- device-side building blocks shared by the gfx950 kernels.
//
// Conventions (see DESIGN.md "Data layout"):
//  * A workgroup owns a tile of S independent streams and runs a whole network stage with the
//    activations in LDS as [row = (time, stream)][channel] matrices.
//  * fp32 activations are stored in "AT16" channel order: inside every aligned block of 16
//    channels the 4x4 index matrix is transposed (phys = (k&3)*4 + (k>>2)&3).  A lane of
//    v_mfma_f32_16x16x4_f32 holds A[m = lane&15][k = lane>>4]; with AT16 one ds_read_b128 returns
//    the lane's A operands for four consecutive MFMAs (k = q, 4+q, 8+q, 12+q ... i.e. kk*4+q), so the
//    K loop runs in strictly ascending k -- bitwise the fmaf chain the oracle defines
//    (oracle/lyra_oracle.c header) -- at one LDS read per four MFMAs.  The chains START FROM THE BIAS
//    (gemm_f32_bias below): that is the order XNNPACK's f32 micro-kernels compute (tests/test_xnnpack_witness.py),
//    and it removes the epilogues' "+ bias" vector add.
//  * fp32 rows are padded by 8 floats (stride C+8): conflict-free for that ds_read_b128 pattern.
//  * int8 activations keep natural channel order, row stride C+32 bytes.
//  * Weights are pre-packed on the host into per-lane MFMA B fragments (one 16-byte global load
//    per lane per K chunk, fully coalesced).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// LYRA_SYNC_KEEP(): the barrier behind which the stream ids / ring phases sit (addresses depend on it).
// -DLYRA_NO_BARRIER is a TIMING-ONLY ablation (results are garbage): every OTHER workgroup barrier becomes a wave barrier --
// how much of a stage kernel is waiting for its slowest wave?
#define LYRA_SYNC_KEEP() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#ifdef LYRA_NO_BARRIER
#define __syncthreads() __builtin_amdgcn_wave_barrier()
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define LYRA_LRELU_ALPHA 0.30000001192092896f

#ifdef LYRA_TIMING
static __device__ long long g_lyra_tdbg[128];
// g_lyra_exit_at = i: every workgroup of this translation unit's kernels returns at stamp i (top-level stamps only) --
// the kernel's duration is then the cumulative cost of the phases before it, under the real co-residency.
static __device__ int g_lyra_exit_at = -1;
#define LYRA_TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lyra_tdbg[i] = clock64(); \
    if (g_lyra_exit_at == (i)) return; } while (0)
#define LYRA_WSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lyra_tdbg[i] = wall_clock64(); } while (0)
#define LYRA_TSTAMP2(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lyra_tdbg[i] = clock64(); } while (0)   // no early exit
// per-workgroup trace: [wg][0] start (100 MHz wall clock), [1] end, [2] HW_ID, [3] XCC_ID
static __device__ long long g_lyra_wgtrace[2048 * 4];
#define LYRA_WG_BEGIN() do { if (threadIdx.x == 0 && blockIdx.x < 2048) { \
    g_lyra_wgtrace[blockIdx.x * 4 + 0] = wall_clock64(); \
    g_lyra_wgtrace[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492); \
    g_lyra_wgtrace[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(63508); } } while (0)
#define LYRA_WG_END() do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_lyra_wgtrace[blockIdx.x * 4 + 1] = wall_clock64(); } while (0)
#else
#define LYRA_WG_BEGIN() do { } while (0)
#define LYRA_WG_END() do { } while (0)
#define LYRA_WSTAMP(i) do { } while (0)
#define LYRA_TSTAMP(i) do { } while (0)
#define LYRA_TSTAMP2(i) do { } while (0)
#endif

// -DLYRA_STRESS_DELAY=mask: ORDERING STRESS build (results must not change): kernel family `bit` sleeps ~0.2 ms at its start
// (0-2 encoder stages, 3 quantizer, 4-6 decoder stages, 7 log-mel / noise estimator, 8 resampler, 9 comfort noise, 10 twin
// helpers, 11 rvq_decode), which moves every kernel of the family against the other streams -- a missing cross-stream
// edge then shows in the parity / fuzz tests (profiles/r06_stress_matrix.txt).
#ifdef LYRA_STRESS_DELAY
#define LYRA_STRESS(bit) do { if (((LYRA_STRESS_DELAY) >> (bit)) & 1) for (int i_ = 0; i_ < 60; ++i_) __builtin_amdgcn_s_sleep(127); } while (0)
#else
#define LYRA_STRESS(bit) do { } while (0)
#endif

namespace lyra {

__host__ __device__ constexpr int at16(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }

// x > 0 ? x : alpha * x  ==  max(x, alpha * x) for 0 < alpha < 1, bit for bit (signed zeros included; activations are never
// NaN): one multiply + one v_max_f32 instead of multiply + compare + select -- on this part every vector instruction is
// issue time that does not hide under the MFMAs, and the float LeakyReLU is a fifth of the fp32 stages'
// vector work.  Spelled as an instruction: fmaxf() would add a canonicalisation of x in IEEE mode.
__device__ __forceinline__ float lrelu(float x) {
  const float ax = x * LYRA_LRELU_ALPHA;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(ax));
  return r;
}
// Four at once: the two multiplies as v_pk_mul_f32 (two fp32 products per issue slot -- the same IEEE products), the
// maxima stay scalar (there is no packed fp32 max): 6 instead of 8 vector instructions per four elements.  Round 6,
// alternating on one box (profiles/r06_ab_pk_lrelu.txt): driver form 14.05 -> 14.09 M, sustained 14.45 -> 14.47 M frames/s
// (-DLYRA_SCALAR_LRELU builds the old form).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 lrelu4(f32x4 v) {
  f32x4 r;
#ifndef LYRA_SCALAR_LRELU
  const f32x2 al = (f32x2){LYRA_LRELU_ALPHA, LYRA_LRELU_ALPHA};
  f32x2 lo = (f32x2){v[0], v[1]}, hi = (f32x2){v[2], v[3]}, alo, ahi;
  // The products are COMPILER-VISIBLE vector multiplies (selected as v_pk_mul_f32): v is usually a fresh MFMA result, and
  // this part does not interlock a vector read against an MFMA still writing its destination -- the compiler knows the
  // wait states of its own instructions, not of an asm block (tools/hazard_probe.hip: a v_pk_mul_f32 spelled in asm right
  // behind the MFMA reads elements 2, 3 stale).  The maxima below read v only after the product of the same registers.
  alo = lo * al;
  ahi = hi * al;
  asm("v_max_f32 %0, %1, %2" : "=v"(r[0]) : "v"(v[0]), "v"(alo[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[1]) : "v"(v[1]), "v"(alo[1]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[2]) : "v"(v[2]), "v"(ahi[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[3]) : "v"(v[3]), "v"(ahi[1]));
#else
  r[0] = lrelu(v[0]); r[1] = lrelu(v[1]); r[2] = lrelu(v[2]); r[3] = lrelu(v[3]);
#endif
  return r;
}
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) {
  f32x4 r;
  r[0] = __builtin_fmaf(a[0], b[0], c[0]); r[1] = __builtin_fmaf(a[1], b[1], c[1]);
  r[2] = __builtin_fmaf(a[2], b[2], c[2]); r[3] = __builtin_fmaf(a[3], b[3], c[3]);
  return r;
}

// ---------------------------------------------------------------------------------------------
// fixed point (TFLite common.h semantics; same formulas as oracle/lyra_oracle.c)
// ---------------------------------------------------------------------------------------------
// SaturatingRoundingDoublingHighMul: gemmlowp nudges by +2^30 (ab >= 0) or 1-2^30 (ab < 0) and divides by 2^31
// truncating toward zero -- which is exactly floor((ab + 2^30) / 2^31), i.e. round-half-up, for either sign.
// One v_mad_i64_i32 for the nudged product and one v_alignbit_b32 for bits [62:31] (round 2 did this on the 32-bit
// halves with a carry: eight instructions).  `b` is a quantized multiplier in [2^30, 2^31) wherever this is called
// (model.hip builds them with QuantizeMultiplier), so gemmlowp's one saturating case, a == b == INT32_MIN, cannot occur.
__device__ __forceinline__ int32_t srdhm(int32_t a, int32_t b) {
  const int64_t p = (int64_t)a * (int64_t)b + (1ll << 30);
  return (int32_t)(p >> 31);
}
__device__ __forceinline__ int32_t rdivpot(int32_t x, int e) {
  const int32_t mask = (int32_t)((1u << e) - 1u);
  const int32_t rem = x & mask;
  const int32_t thr = (mask >> 1) + (x < 0 ? 1 : 0);
  return (x >> e) + (rem > thr ? 1 : 0);
}
__device__ __forceinline__ int32_t mbqm_double(int32_t x, int32_t M, int shift) {
  const int left = shift > 0 ? shift : 0;
  const int right = shift > 0 ? 0 : -shift;
  return rdivpot(srdhm((int32_t)((uint32_t)x << left), M), right);
}
__device__ __forceinline__ int32_t mbqm_exact(int32_t x, int32_t M, int shift) {
  int total = 31 - shift;
  return (int32_t)(((int64_t)x * (int64_t)M + (1ll << (total - 1))) >> total);
}
// conv / depthwise / transpose-conv requantisation: mode 0 exact, 1 gemmlowp double rounding
__device__ __forceinline__ int32_t requant(int32_t acc, int32_t M, int shift, int mode) {
  return mode ? mbqm_double(acc, M, shift) : mbqm_exact(acc, M, shift);
}
__device__ __forceinline__ int32_t clamp8(int32_t v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

// TFLite AffineQuantize: round-half-away(x / s) + z, clamped -- bitwise the oracle's quantize_f (roundf(x / s) with an
// IEEE division), WITHOUT the division: an IEEE fp32 division is a ~14-instruction sequence on this ISA and the int8
// stages quantize 24 (dec_s0) / 16 (enc_s2) activations per thread, 14 % / 16 % of all their VALU work.  With
// rs = RN(1 / s) from the host (QP::rs), q1 = RN(x * rs) is within one ulp of x / s, the remainder x - q1 * s is exact
// in one fma, and RN(q1 + rem * rs) is the correctly rounded quotient (Markstein's correction step).  x is first
// clamped to +-lim = +-512 * s: beyond that the result is the clamp either way, and no intermediate can overflow.
// Proved, not assumed: oracle/quantize_proof.c runs this sequence against roundf(x / s) over ALL 2^32 float inputs for
// every scale the two graphs quantize with (tests/test_quantize_division_free.py).
struct QP { float s; int32_t z; float rs, lim; };
__device__ __forceinline__ int32_t quantize_f(float x, const QP& Q) {
  x = __builtin_fminf(__builtin_fmaxf(x, -Q.lim), Q.lim);
  const float q1 = x * Q.rs;
  const float rem = __builtin_fmaf(-q1, Q.s, x);
  const float q = __builtin_fmaf(rem, Q.rs, q1);
  return clamp8((int32_t)__builtin_roundf(q) + Q.z);
}
// float(double(s) * (q - z)) == s * float(q - z) in fp32 (24-bit x 9-bit product is exact before the one rounding)
__device__ __forceinline__ float dequantize_f(int32_t q, float s, int32_t z) { return s * (float)(q - z); }
__device__ __forceinline__ float dequantize_f(int32_t q, const QP& Q) { return dequantize_f(q, Q.s, Q.z); }

struct LreluQ { int32_t zin, zout, mpos, spos, mneg, sneg; };
struct AddQ { int32_t z1, z2, zo, m1, s1, m2, s2, mo, so; };

__device__ __forceinline__ int32_t lrelu_q(int32_t x, const LreluQ& L) {
  int32_t v = x - L.zin;
  int32_t r = v >= 0 ? mbqm_double(v, L.mpos, L.spos) : mbqm_double(v, L.mneg, L.sneg);
  return clamp8(r + L.zout);
}
__device__ __forceinline__ int32_t add_q(int32_t a, int32_t b, const AddQ& L) {
  int32_t va = (a - L.z1) * (1 << 20);
  int32_t vb = (b - L.z2) * (1 << 20);
  int32_t sa = mbqm_double(va, L.m1, L.s1);
  int32_t sb = mbqm_double(vb, L.m2, L.s2);
  return clamp8(mbqm_double(sa + sb, L.mo, L.so) + L.zo);
}

// ---------------------------------------------------------------------------------------------
// MODE 2 "xnnpack": XNNPACK's QS8 arithmetic (what the reference runs, use_xnn=true; every formula held against real
// XNNPACK code, tests/test_xnnpack_witness.py).  Cheaper on this ISA than TFLite's fixed-point emulation as well.
// ---------------------------------------------------------------------------------------------
// conv / depthwise / transpose-conv: q = RNE(clamp(float(acc) * scale)) + z.  `Mbits` carries the fp32 scale
// (s_in * s_w[c]) / s_out in mode 2 and the Q31 multiplier otherwise (model.hip).  The clamp bounds are integers, so
// clamp-then-round == round-then-clamp; v_rndne_f32 is round-to-nearest-even like lrintf / cvtps2dq.
// Rounding by the magic number 1.5 * 2^23: for |v| < 2^22 the float sum v + 12582912.f is RNE(v) + 12582912 exactly (one
// rounding, ties to even, like v_rndne_f32 / lrintf), and its bit pattern is 0x4B400000 + RNE(v).  v is clamped first (the
// bounds are integers, so clamp-then-round == round-then-clamp): cvt, mul, med3, add, sub -- and when a table lookup
// follows, the sub folds into the table's base address.
__device__ __forceinline__ int32_t rne_clamped_code(float v, int32_t z) {
  const float lo = (float)(-128 - z), hi = (float)(127 - z);
  const float c = __builtin_amdgcn_fmed3f(v, lo, hi);   // one v_med3_f32 (v is never NaN: finite operands)
  return __float_as_int(c + 12582912.f) - (0x4B400000 - z);
}
__device__ __forceinline__ int32_t xnn_requant(int32_t acc, int32_t Mbits, int32_t zout) {
  return rne_clamped_code((float)acc * __int_as_float(Mbits), zout);
}
// Mode 3 "builtin_mixed" (round 6): what the graphs compute if TFLite's XNNPACK delegate takes the fp32 operators but NOT
// the signed-int8 ones (the reference sets only the QU8 delegate flag, tflite_model_wrapper.cc:65-67) -- TFLite 2.11's builtin
// int8 kernels per operator, as recalled in DESIGN.md 2: an UNGROUPED CONV_2D runs on the optimized path (single rounding =
// flavour 0), a grouped CONV_2D, DEPTHWISE_CONV_2D and TRANSPOSE_CONV on the reference kernels (gemmlowp double rounding =
// flavour 1); LEAKY_RELU / ADD / QUANTIZE are the builtin forms modes 0 and 1 share.  conv_flavour<MODE, UNGROUPED>() is the
// conv_code / conv_dequant template argument of ONE layer: the kernel's mode everywhere but in mode 3.
template <int MODE, bool UNGROUPED_CONV>
__host__ __device__ constexpr int conv_flavour() { return MODE == 3 ? (UNGROUPED_CONV ? 0 : 1) : MODE; }
// the layers' requantisation in flavour 0 / 1 / 2 -> clamped int8 code (as int)
template <int MODE>
__device__ __forceinline__ int32_t conv_code(int32_t acc, int32_t M, int sh, int32_t zout) {
  static_assert(MODE >= 0 && MODE <= 2, "a layer's flavour (conv_flavour<>), not the kernel's mode");
  if constexpr (MODE == 2) return xnn_requant(acc, M, zout);
  else return clamp8(requant(acc, M, sh, MODE) + zout);
}
// requantise and dequantise at once (a conv output that a DEQUANTIZE of the SAME tensor follows -- same zero point, model.hip
// checks --: the graphs' int8 ->
// fp32 hand-overs): in mode 2 the rounded, clamped value is already a float -- (c + magic) - magic, exact -- so the code
// never has to exist as an integer: s * float(code - z) == s * RNE(clamp(v)).
template <int MODE>
__device__ __forceinline__ float conv_dequant(int32_t acc, int32_t M, int sh, int32_t zout, float s_dq) {
  if constexpr (MODE == 2) {
    const float lo = (float)(-128 - zout), hi = (float)(127 - zout);
    const float c = __builtin_amdgcn_fmed3f((float)acc * __int_as_float(M), lo, hi);
    const float r = (c + 12582912.f) - 12582912.f;          // RNE(c) = code - zout, still a float
    return s_dq * r;
  } else {
    return dequantize_f(conv_code<MODE>(acc, M, sh, zout), s_dq, zout);
  }
}
// f32 -> qs8 convert: RNE(x * (1 / s)) + z, clamped (reciprocal multiply, ties to even)
__device__ __forceinline__ int32_t xnn_quantize(float x, const QP& Q) { return rne_clamped_code(x * Q.rs, Q.z); }
template <int MODE>
__device__ __forceinline__ int32_t quantize_code(float x, const QP& Q) {
  if constexpr (MODE == 2) return xnn_quantize(x, Q);
  else return quantize_f(x, Q);
}
// qs8 add: (a * ma + b * mb + bias) >> shift, clamped around the zero point.  AddQ in mode 2 (model.hip):
// m1 = ma, m2 = mb, mo = bias (rounding term and both zero points folded in), so = shift.
__device__ __forceinline__ int32_t xnn_add(int32_t a, int32_t b, const AddQ& L) {
  const int32_t acc = L.mo + a * L.m1 + b * L.m2;
  return clamp8((acc >> L.so) + L.zo);
}

// ---------------------------------------------------------------------------------------------
// MFMA tile GEMMs: A from LDS, B fragments from global (L2-resident weights), C in registers.
//   a_off(i, c)  -> offset (floats / bytes) of this lane's 16-byte A fragment for the wave's i-th
//                   M tile and K chunk c (lane-specific part included by the caller's lambda)
//   bfrag        -> this wave's first N tile; tile j at bfrag + j * KC * 64
// Accumulators are returned to the caller (epilogues often straddle a barrier).
// ---------------------------------------------------------------------------------------------
// The weight pointers come out of a parameter block in memory, so the compiler only knows them as generic
// (flat) pointers; say they are global so the loads are global_load (vmcnt only, no LDS aperture check).
#define LYRA_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const T LYRA_GLOBAL* as_global(const T* p) {
  return (const T LYRA_GLOBAL*)p;
}
// A pointer every lane of the wave holds the same value of (a weight fragment base: it depends on the wave's index and on
// kernel parameters only), moved to SGPRs: the loads through it then take the scalar-base form
// `global_load ..., v_lane_offset, s[base:base+1]`, and stepping from K chunk to K chunk is scalar arithmetic instead of a
// 64-bit vector add per load (the chunks are 1 KB apart, beyond the instruction's immediate offset after four of them).
// base (wave-uniform: a kernel argument advanced by a tile offset) + a 32-bit per-lane BYTE offset, as a global pointer:
// the access takes the scalar-base form and the lane arithmetic stays 32-bit.  Every buffer of the path is < 4 GB.
template <class T, class U>
__device__ __forceinline__ T LYRA_GLOBAL* goff(U* base, uint32_t byte_off) {
  return (T LYRA_GLOBAL*)((uint8_t LYRA_GLOBAL*)base + byte_off);
}
template <class T>
__device__ __forceinline__ const T LYRA_GLOBAL* wave_uniform(const T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const T LYRA_GLOBAL*)(((uint64_t)hi << 32) | lo);
}

// Experiments on how co-resident workgroups share a SIMD (build flags, see DESIGN.md "what was tried"):
//   LYRA_PRIO_MFMA   raise the wave's priority around every MFMA cluster (guide T5)
//   LYRA_PRIO_SLOT   static priority = the wave's slot on its SIMD (0..3): co-resident tiles pipeline instead of
//                    marching in lock-step
//   LYRA_STAGGER=N   tiles in wave slot k start k * N * 64 cycles late
#ifdef LYRA_PRIO_MFMA
#define LYRA_MFMA_BEGIN() __builtin_amdgcn_s_setprio(2)
#define LYRA_MFMA_END() __builtin_amdgcn_s_setprio(0)
#else
#define LYRA_MFMA_BEGIN() do { } while (0)
#define LYRA_MFMA_END() do { } while (0)
#endif
__device__ __forceinline__ void wg_schedule_hint() {
#if defined(LYRA_PRIO_SLOT) || defined(LYRA_STAGGER) || defined(LYRA_STAGGER2)
  const unsigned slot = __builtin_amdgcn_s_getreg(63492) & 15u;   // HW_ID.wave_id: the wave's slot on its SIMD
#endif
#ifdef LYRA_PRIO_SLOT
  switch (slot & 3u) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
#endif
#ifdef LYRA_STAGGER
  for (unsigned i = 0; i < (slot & 3u) * LYRA_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
#endif
#ifdef LYRA_STAGGER2   // half of a CU's co-resident tiles (wave slots 2, 3 of each SIMD) start LYRA_STAGGER2 * 64 cycles late
  if (slot & 2u) __builtin_amdgcn_s_sleep(LYRA_STAGGER2);
#endif
}

// Software-pipelined: the B fragments (L2, ~500+ cycles) and A fragments (LDS) of chunk c+PF are requested
// before the MFMAs of chunk c issue; the K loop is fully unrolled so the PF+1 register stages are static and
// the compiler emits counted s_waitcnt (the prefetches stay in flight across the MFMA block).
//   KS   = K-chunk stride between a tile's fragments in memory (> KC when only part of the packed K range is run)
//   ZERO = start the chains from 0; otherwise acc carries values in (a chain continued from an earlier GEMM)
// LYRA_WEIGHT_ALIAS (timing experiment only, results are wrong): every K chunk of a tile re-reads the tile's first
// weight fragment -- the GEMM's L2 -> CU weight traffic collapses to one 1 KB fragment per wave and N tile while the
// instruction stream stays the same.  What that buys is what weight bandwidth costs.
#ifdef LYRA_WEIGHT_ALIAS
#define LYRA_WCHUNK(c) 0
#else
#define LYRA_WCHUNK(c) (c)
#endif
// INIT: 0 = the chains start from 0, 1 = acc carries values in (a chain continued from an earlier GEMM), 2 = the chains
// start from init[j] (the bias splat of N tile j): the first MFMA of every chain takes it as its C operand directly,
// so no accumulator is written (or even allocated) before the first products arrive.
// The first PF weight chunks (and the bias splats) of a GEMM, requested by gemm_f32_wprefetch BEFORE the barrier / elementwise
// phase that precedes the GEMM: an L2 round trip (500+ cycles, more under load) per GEMM otherwise stands between the barrier
// and the first MFMA -- hidden by the other resident tiles at 4,096 streams, fully exposed with one tile per CU.
template <int NTW, int PF>
struct WPre { f32x4 b[PF][NTW]; f32x4 init[NTW]; };

// SWAP (round 6, the 64-channel stages): the weight fragment is the MFMA's A operand and the activation fragment its B
// operand -- both have the same per-lane shape, so the LDS reads and the packed fragments stay as they are -- which
// transposes the C tile: lane l holds OUTPUT CHANNELS 4 * (l >> 4) + e (e = 0..3) of N tile j for activation row l & 15
// of M tile i.  With the weight columns of a tile packed in AT16 order (model.hip: MFMA row r carries the logical channel
// whose physical position is r) those four channels are four CONSECUTIVE floats of the row: an epilogue is one
// ds_write_b128 / global dwordx4 store per C tile instead of four scalar stores to rows 4 apart (two of which always
// shared an LDS bank at a row stride of 72 floats), the stage input of dec_s2 one dwordx4 load instead of four dword
// loads.  Each output element is still bias, then fma over ascending k: the products commute, the chain does not change.
// The bias comes as the lane's four channels (AT16-ordered array, one 16-byte load).
#ifndef LYRA_SWAP64
#define LYRA_SWAP64 1
#endif
// Measured, alternating on one box (profiles/r06_ab_swap64.txt, r06_ab_swap128.txt): the 64-channel stages swapped
// 14.08 -> 14.13 M frames/s in the driver form and 14.43 -> 14.51 M sustained (static vector instructions of enc_s0 735 ->
// 636, LDS instructions 180 -> 159); the 128-channel stages swapped as well LOSE 0.3 % again (their X update becomes a
// 16-byte read-modify-write per C tile, and dec_s1 is at the register cap): -DLYRA_SWAP128=1 builds that, the default is off.
#ifndef LYRA_SWAP128   // ... and the residual blocks + strided conv of the 128-channel stages
#define LYRA_SWAP128 0
#endif
// this lane's four bias values of the N tile whose first channel is n0 (bias array in AT16 order)
__device__ __forceinline__ f32x4 bias_quad(const float* bias, int n0) {
  return *reinterpret_cast<const f32x4 LYRA_GLOBAL*>(as_global(bias) + n0 + (((threadIdx.x & 63) >> 4) << 2));
}

template <int NTW, int KC, int KS, int PF, bool SWAP = false>
__device__ __forceinline__ WPre<NTW, PF> gemm_f32_wprefetch(const f32x4* bfrag_generic, const float* bias, int n0) {
  const int lane = threadIdx.x & 63;
  const f32x4 LYRA_GLOBAL* bbase = wave_uniform(bfrag_generic);
  WPre<NTW, PF> pre;
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int j = 0; j < NTW; ++j) pre.b[p][j] = bbase[(j * KS + LYRA_WCHUNK(p < KC ? p : 0)) * 64 + lane];
  if constexpr (SWAP) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) pre.init[j] = bias_quad(bias, n0 + j * 16);
  } else {
    const float LYRA_GLOBAL* b = as_global(bias) + n0 + (threadIdx.x & 15);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float v = b[j * 16];
      pre.init[j] = (f32x4){v, v, v, v};
    }
  }
  return pre;
}

template <int MTW, int NTW, int KC, int KS, int INIT, int PF, bool SWAP = false, class AOff>
__device__ __forceinline__ void gemm_f32_core(const float* lds, AOff a_off, const f32x4* bfrag_generic,
                                              f32x4 (&acc)[MTW][NTW], const f32x4 (&init)[NTW],
                                              const f32x4 (*pre_b)[NTW] = nullptr) {
  const int lane = threadIdx.x & 63;
  const f32x4 LYRA_GLOBAL* bbase = wave_uniform(bfrag_generic);
  if (INIT == 2) {   // the splat lives in the LAST M tile's accumulator: the first MFMA of tile i reads it from there and
                     // tile MTW - 1 (issued last) overwrites it in place -- no register beyond the accumulators themselves
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[MTW - 1][j] = init[j];
  }
  f32x4 bq[PF + 1][NTW], aq[PF + 1][MTW];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < KC) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) bq[p][j] = pre_b ? pre_b[p][j] : bbase[(j * KS + LYRA_WCHUNK(p)) * 64 + lane];
#pragma unroll
      for (int i = 0; i < MTW; ++i) aq[p][i] = *reinterpret_cast<const f32x4*>(lds + a_off(i, p));
    }
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    if (c + PF < KC) {
      const int sl = (c + PF) % (PF + 1);
#pragma unroll
      for (int j = 0; j < NTW; ++j) bq[sl][j] = bbase[(j * KS + LYRA_WCHUNK(c + PF)) * 64 + lane];
#pragma unroll
      for (int i = 0; i < MTW; ++i) aq[sl][i] = *reinterpret_cast<const f32x4*>(lds + a_off(i, c + PF));
    }
#ifdef LYRA_GEMM_SCHED_BARRIER   // experiment: keep the prefetch loads above this chunk's MFMAs (the scheduler sinks them otherwise)
    __builtin_amdgcn_sched_barrier(0);
#endif
    const int cur = c % (PF + 1);
    LYRA_MFMA_BEGIN();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const bool first = c == 0 && kk == 0;
          const f32x4 cin = (first && INIT == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : (first && INIT == 2) ? acc[MTW - 1][j] : acc[i][j];
          acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(bq[cur][j][kk], aq[cur][i][kk], cin, 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cur][i][kk], bq[cur][j][kk], cin, 0, 0, 0);
        }
    LYRA_MFMA_END();
  }
}

template <int MTW, int NTW, int KC, int KS = KC, bool ZERO = true,
          int PF = (4 * MTW * NTW >= 16 ? 1 : (4 * MTW * NTW >= 8 ? 2 : 3)), class AOff>
__device__ __forceinline__ void gemm_f32(const float* lds, AOff a_off, const f32x4* bfrag_generic,
                                         f32x4 (&acc)[MTW][NTW]) {
  f32x4 none[NTW];
  gemm_f32_core<MTW, NTW, KC, KS, ZERO ? 0 : 1, PF>(lds, a_off, bfrag_generic, acc, none);
}

// The chains start from the bias (logical channel order; n0 = first output channel of the wave's first N tile):
// C layout col = lane & 15, so one value per N tile fills a lane's four rows.
template <int MTW, int NTW, int KC, int KS = KC,
          int PF = (4 * MTW * NTW >= 16 ? 1 : (4 * MTW * NTW >= 8 ? 2 : 3)), bool SWAP = false, class AOff>
__device__ __forceinline__ void gemm_f32_bias(const float* lds, AOff a_off, const f32x4* bfrag_generic,
                                              const float* bias, int n0, f32x4 (&acc)[MTW][NTW]) {
  f32x4 init[NTW];
  if constexpr (SWAP) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) init[j] = bias_quad(bias, n0 + j * 16);
  } else {
    const float LYRA_GLOBAL* b = as_global(bias) + n0 + (threadIdx.x & 15);
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const float v = b[j * 16];
      init[j] = (f32x4){v, v, v, v};
    }
  }
  gemm_f32_core<MTW, NTW, KC, KS, 2, PF, SWAP>(lds, a_off, bfrag_generic, acc, init);
}

// ... with the first weight chunks and the bias splats requested earlier (gemm_f32_wprefetch with the same NTW, KC, KS, PF)
template <int MTW, int NTW, int KC, int KS = KC,
          int PF = (4 * MTW * NTW >= 16 ? 1 : (4 * MTW * NTW >= 8 ? 2 : 3)), bool SWAP = false, class AOff>
__device__ __forceinline__ void gemm_f32_pre(const float* lds, AOff a_off, const f32x4* bfrag_generic,
                                             const WPre<NTW, PF>& pre, f32x4 (&acc)[MTW][NTW]) {
  gemm_f32_core<MTW, NTW, KC, KS, 2, PF, SWAP>(lds, a_off, bfrag_generic, acc, pre.init, pre.b);
}
template <int MTW, int NTW>
constexpr int gemm_pf() { return 4 * MTW * NTW >= 16 ? 1 : (4 * MTW * NTW >= 8 ? 2 : 3); }

// ... from splats the caller already holds
template <int MTW, int NTW, int KC, int KS = KC,
          int PF = (4 * MTW * NTW >= 16 ? 1 : (4 * MTW * NTW >= 8 ? 2 : 3)), bool SWAP = false, class AOff>
__device__ __forceinline__ void gemm_f32_init(const float* lds, AOff a_off, const f32x4* bfrag_generic,
                                              const f32x4 (&init)[NTW], f32x4 (&acc)[MTW][NTW]) {
  gemm_f32_core<MTW, NTW, KC, KS, 2, PF, SWAP>(lds, a_off, bfrag_generic, acc, init);
}

template <int MTW, int NTW, int KC, class AOff>
__device__ __forceinline__ void gemm_i8(const int8_t* lds, AOff a_off, const i32x4* bfrag_generic,
                                        i32x4 (&acc)[MTW][NTW]) {
  const int lane = threadIdx.x & 63;
  const i32x4 LYRA_GLOBAL* bfrag = wave_uniform(bfrag_generic);
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    i32x4 b[NTW], a[MTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) b[j] = bfrag[(j * KC + c) * 64 + lane];
#pragma unroll
    for (int i = 0; i < MTW; ++i) a[i] = *reinterpret_cast<const i32x4*>(lds + a_off(i, c));
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// Operand-swapped int8 GEMM: the weight fragment is the A operand and the activation fragment the B operand (both have
// the same per-lane shape -- 16 consecutive k of one row / column --, so nothing is repacked), which transposes the C
// tile: lane l holds OUT CHANNELS 4 * (l >> 4) + e (e = 0..3) of the wave's N tile j for activation row l & 15.  An
// epilogue then packs its four codes into ONE dword store (the un-swapped layout scatters them over four rows: four
// ds_write_b8 at stride QS, the bank-conflict hot spot of rounds 2-3), reads a residual row's four bytes with one dword
// load, and takes its per-channel parameters as 16-byte quads.  acc comes in initialised (bias - zin * sum(w)).
template <int NTW, int KC, class AOff>
__device__ __forceinline__ void gemm_i8_t(const int8_t* lds, AOff a_off, const i32x4* bfrag_generic, i32x4 (&acc)[NTW]) {
  const int lane = threadIdx.x & 63;
  const i32x4 LYRA_GLOBAL* bfrag = wave_uniform(bfrag_generic);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    i32x4 w[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) w[j] = bfrag[(j * KC + c) * 64 + lane];
    const i32x4 a = *reinterpret_cast<const i32x4*>(lds + a_off(c));
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[j], a, acc[j], 0, 0, 0);
  }
}
// this lane's four consecutive per-channel parameters of N tile nt (swapped layout)
__device__ __forceinline__ i32x4 chan_quad(const int32_t* p, int nt) {
  return *reinterpret_cast<const i32x4 LYRA_GLOBAL*>(as_global(p) + nt * 16 + (((threadIdx.x & 63) >> 4) << 2));
}

// A 16-row tile whose rows 8..15 are padding (GEMM rows = 8 streams) leaves lanes 32-63 idle in the epilogue.
// fold_rows8() moves the valid rows of tile j + H into the upper lanes of tile j (v_permlane32_swap: upper half of
// the first operand <-> lower half of the second): afterwards lane L holds in acc[j] (j < H) the accumulators of
// tile j + H * (L >> 5), row ((L >> 4) & 1) * 4 + reg, and every lane carries valid rows.
template <int H>
__device__ __forceinline__ void fold_rows8(i32x4 (&acc)[2 * H]) {
#pragma unroll
  for (int j = 0; j < H; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc[j][e] = (int)__builtin_amdgcn_permlane32_swap((unsigned)acc[j][e], (unsigned)acc[j + H][e], false, false)[0];
}
template <int H>
__device__ __forceinline__ void fold_rows8(f32x4 (&acc)[2 * H]) {
#pragma unroll
  for (int j = 0; j < H; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc[j][e] = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][e]),
                                                                   __float_as_uint(acc[j + H][e]), false, false)[0]);
}

// Rotates two registers by 32 lanes (rows +-8 of a 16-row C tile): a <- [a.hi, a.lo], b <- [b.hi, b.lo].
__device__ __forceinline__ void rot32_pair(float& a, float& b) {
  auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);  // [a.lo,b.lo],[a.hi,b.hi]
  auto r2 = __builtin_amdgcn_permlane32_swap(r1[1], r1[0], false, false);                            // [a.hi,a.lo],[b.hi,b.lo]
  a = __uint_as_float(r2[0]);
  b = __uint_as_float(r2[1]);
}

// C/D layout of every 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
__device__ __forceinline__ int cd_col() { return threadIdx.x & 15; }
__device__ __forceinline__ int cd_row(int reg) { return (((threadIdx.x & 63) >> 4) << 2) + reg; }

// ---------------------------------------------------------------------------------------------
// parameter blocks (device pointers into the context's weight arena)
// ---------------------------------------------------------------------------------------------
struct ConvF { const f32x4* w; const float* b; };          // b: logical channel order
struct DwF { const float* w; const float* b; };            // [3][C], [C] in AT16 channel order
struct ConvQ { const i32x4* w; const int32_t* b; const int32_t* M; const int32_t* sh; int32_t zout; };
//   b already holds bias - zin * sum_k(w): the GEMM runs on raw int8 codes
//   DwQ::b likewise holds bias - zin * sum_j(w[j][c])
struct DwQ { const int8_t* w; const int32_t* b; const int32_t* M; const int32_t* sh; int32_t zin, zout; };

// ---------------------------------------------------------------------------------------------
// L2 / TLB warm-up.  Kernel boundaries invalidate the XCD L2s, so every kernel starts with its weights cold
// and -- all workgroups marching through the same layer sequence in lockstep -- each layer's first weight
// fetch would be a compulsory miss that every workgroup waits on, in the middle of the dependent phase chain.
// Instead the workgroups of an XCD (blockIdx round-robins over the 8 XCDs) share one early pass touching
// every 128-byte line of the kernel's weight range; by the time the layers run, their weights are L2 hits.
// Returns a value the caller must consume at the END of the kernel (l2_warm_sink) so the loads stay in flight.
// ---------------------------------------------------------------------------------------------
struct WarmRange { const uint8_t* base; uint32_t bytes; };

template <int MAXIT>
struct WarmTok { uint32_t v[MAXIT]; };

template <int NT, int MAXIT>
__device__ __forceinline__ WarmTok<MAXIT> l2_warm(const WarmRange& w) {
  const uint32_t lines = w.bytes >> 7;
  const uint32_t wgs = (gridDim.x + 7) >> 3, me = blockIdx.x >> 3;
  const uint32_t per = (lines + wgs - 1) / wgs;
  const uint32_t lo = me * per;
  uint32_t hi = lo + per;
  hi = hi < lines ? hi : lines;
  WarmTok<MAXIT> tok;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const uint32_t l = lo + it * NT + threadIdx.x;
    tok.v[it] = 0;
    if (l < hi) tok.v[it] = *reinterpret_cast<const uint32_t LYRA_GLOBAL*>(as_global(w.base) + ((size_t)l << 7));
  }
  return tok;
}
// The kernel's OWN machine code, same idea.  The stage kernels are tens of KB of straight-line code executed once
// per workgroup; instruction-cache misses are served by the XCD's L2, which is cold at a kernel boundary, so each
// line's first fetch goes to the Infinity Cache -- or, when the step's working set (265 MB of per-stream state at
// B = 4096, more than the 256 MB cache) has pushed the code out of it, to HBM, in the middle of the dependent phase
// chain.  Measured (pipeline_probe.py (a probe of an earlier round, removed since: git history)): inside the sustained encode+decode pipeline the two largest kernels
// (enc_s2 29 KB, dec_s0 42 KB of code) ran 64 / 82 us instead of 40 / 45 us on most boxes of the pool; pulling
// the code into L2 with data loads at kernel start brings them back to 51 / 61 us.
// `code_bytes` comes from the host: the symbol size of this kernel minus the offset of this very s_getpc_b64 inside it,
// both read back from the kernel object at build time (code_sizes.sh) -- the range cannot run past the function.
template <int NT>
__device__ __forceinline__ WarmTok<1> code_warm(int code_bytes) {
  const uint64_t pc = __builtin_amdgcn_s_getpc();
  return l2_warm<NT, 1>(WarmRange{reinterpret_cast<const uint8_t*>(pc & ~127ull), (uint32_t)(code_bytes > 0 ? code_bytes : 0)});
}

// never true in practice; keeps the warm-up loads alive without waiting for them early
template <int MAXIT>
__device__ __forceinline__ void l2_warm_sink(const WarmTok<MAXIT>& tok, uint8_t* state, int B) {
  uint32_t acc = 0;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) acc ^= tok.v[it];
  if (acc == 0x9E3779B9u && B == -0x5EED) state[0] = (uint8_t)acc;
}

}  // namespace lyra
