/* quantize_proof.c -- TEST INFRASTRUCTURE (CPU): exhaustive proof that the division-free quantisation sequence of the
 * HIP kernels (lyra_amd/csrc/lyra_dev.h quantize_f) equals TFLite's AffineQuantize as the oracle states it
 * (oracle/lyra_oracle.c quantize_f: clamp(roundf(x / s) + z), IEEE division) for EVERY float input.
 *
 *   lo_quantize_proof(s, z, threads, full, &first_bad) -> number of float bit patterns (NaNs excluded) on which the two
 *   differ.  full = 1: all 2^32 patterns (24 s per scale on 8 cores; run once per model, log in profiles/);
 *   full = 0: every float of either sign with s / 8 <= |x| <= 2048 s -- the 14 binades around the 256 code boundaries;
 *   outside them both sides are trivially 0 + z or the clamp.
 *
 * fmaf / the fp32 multiply here are the IEEE operations v_fma_f32 / v_mul_f32 implement (denormals included), so a
 * clean sweep on the host is a proof for the device sequence.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

static inline int32_t clamp8(int32_t v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

static inline int32_t quantize_ref(float x, float s, int32_t z) {
  float r = roundf(x / s);
  /* (int32_t) of a float beyond int range is undefined in C; the oracle only ever sees activations, the sweep sees
   * everything: saturate the way v_cvt_i32_f32 does */
  int32_t q = r >= 2147483648.f ? INT32_MAX : (r <= -2147483648.f ? INT32_MIN : (int32_t)r);
  int64_t t = (int64_t)q + z;
  return t < -128 ? -128 : (t > 127 ? 127 : (int32_t)t);
}

static inline int32_t quantize_fast(float x, float s, float rs, float lim, int32_t z) {
  x = fminf(fmaxf(x, -lim), lim);
  const float q1 = x * rs;
  const float rem = fmaf(-q1, s, x);
  const float q = fmaf(rem, rs, q1);
  return clamp8((int32_t)roundf(q) + z);
}

typedef struct { float s; int32_t z; uint32_t lo, hi; uint64_t bad; uint32_t first_bad; } job_t;

static void* worker(void* p) {
  job_t* j = (job_t*)p;
  const float s = j->s, rs = 1.f / s, lim = 512.f * s;
  uint64_t bad = 0;
  uint32_t first = 0;
  for (uint64_t u = j->lo; u <= j->hi; ++u) {
    uint32_t b = (uint32_t)u;
    float x;
    memcpy(&x, &b, 4);
    if (x != x) continue;
    if (quantize_ref(x, s, j->z) != quantize_fast(x, s, rs, lim, j->z)) {
      if (!bad) first = b;
      ++bad;
    }
  }
  j->bad = bad;
  j->first_bad = first;
  return 0;
}

static uint64_t sweep(float s, int32_t z, int threads, uint64_t lo, uint64_t hi, uint32_t* first_bad) {
  pthread_t th[64];
  job_t jobs[64];
  const uint64_t total = hi - lo + 1, per = total / (uint64_t)threads;
  for (int t = 0; t < threads; ++t) {
    jobs[t].s = s; jobs[t].z = z;
    jobs[t].lo = (uint32_t)(lo + per * (uint64_t)t);
    jobs[t].hi = (uint32_t)(t == threads - 1 ? hi : lo + per * (uint64_t)(t + 1) - 1);
    pthread_create(&th[t], 0, worker, &jobs[t]);
  }
  uint64_t bad = 0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(th[t], 0);
    if (jobs[t].bad && !bad && first_bad) *first_bad = jobs[t].first_bad;
    bad += jobs[t].bad;
  }
  return bad;
}

uint64_t lo_quantize_proof(float s, int32_t z, int threads, int full, uint32_t* first_bad) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if (full) return sweep(s, z, threads, 0, 0xFFFFFFFFull, first_bad);
  float a = s / 8.f, b = 2048.f * s;
  uint32_t ua, ub;
  memcpy(&ua, &a, 4);
  memcpy(&ub, &b, 4);
  uint64_t bad = sweep(s, z, threads, ua, ub, first_bad);
  if (bad) return bad;
  return sweep(s, z, threads, 0x80000000ull | ua, 0x80000000ull | ub, first_bad);
}
