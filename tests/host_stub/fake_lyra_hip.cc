// fake_lyra_hip.cc -- TEST INFRASTRUCTURE: a CPU stand-in for the handful of C-ABI entry points the C++ plugin layer
// (lyra_amd/host/lyra_hip_components.cc) calls, so that the layer's host logic -- stream slots, call combining,
// argument validation -- runs in the CPU test suite.  Every "device call" sleeps a little (so that concurrent callers
// pile up behind it, as they do behind a real launch), checks that a batch names no stream twice, counts itself, and
// returns outputs that are a pure function of (stream id, input): a combined call must give each caller exactly what
// its own B = 1 call would have.  Never linked into the product.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <set>
#include <thread>

#include "../../include/lyra_hip.h"

struct lyra_hip_ctx { int max_streams; std::atomic<int> in_call{0}; };
static std::atomic<long> g_calls{0}, g_rows{0};
static std::atomic<int> g_overlap{0};

extern "C" long fake_device_calls() { return g_calls.load(); }
extern "C" long fake_device_rows() { return g_rows.load(); }
extern "C" int fake_overlapping_calls() { return g_overlap.load(); }   // calls that ran inside another (must stay 0)

namespace {
struct CallScope {
  lyra_hip_ctx* c;
  CallScope(lyra_hip_ctx* c_, int B) : c(c_) {
    if (c->in_call.fetch_add(1) != 0) g_overlap.fetch_add(1);   // the C ABI wants the calls on ONE context serialised
    g_calls += 1;
    g_rows += B;
    { static const int us = getenv("FAKE_CALL_US") ? atoi(getenv("FAKE_CALL_US")) : 300; if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us)); }
  }
  ~CallScope() { c->in_call.fetch_sub(1); }
};
bool distinct(const int32_t* ids, int B) {
  std::set<int32_t> s(ids, ids + B);
  return (int)s.size() == B;
}
}  // namespace

extern "C" {
int lyra_hip_create(const char*, int, int max_streams, int, lyra_hip_ctx** out) {
  *out = new lyra_hip_ctx; (*out)->max_streams = max_streams;
  return 0;
}
void lyra_hip_destroy(lyra_hip_ctx* c) { delete c; }
const char* lyra_hip_last_error(const lyra_hip_ctx*) { return "fake"; }
int lyra_hip_reset_streams(lyra_hip_ctx*, const int32_t*, int) { return 0; }
int lyra_hip_extract(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, float* feats) {
  CallScope s(c, B);
  if (B > c->max_streams || !distinct(ids, B)) return LYRA_HIP_EINVAL;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 64; ++i) feats[b * 64 + i] = (float)ids[b] * 1000.f + (float)pcm[b * 320 + i] + (float)i * 0.5f;
  return 0;
}
int lyra_hip_logmel(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, float* mel) {
  CallScope s(c, B);
  if (!distinct(ids, B)) return LYRA_HIP_EINVAL;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 160; ++i) mel[b * 160 + i] = (float)ids[b] + (float)pcm[b * 320 + i] * 0.25f;
  return 0;
}
int lyra_hip_generate(lyra_hip_ctx* c, const int32_t* ids, int B, const float* feats, int16_t* pcm) {
  CallScope s(c, B);
  if (!distinct(ids, B)) return LYRA_HIP_EINVAL;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 320; ++i) pcm[b * 320 + i] = (int16_t)((int)feats[b * 64 + (i & 63)] % 1000 + ids[b] + i);
  return 0;
}
int lyra_hip_rvq_encode(lyra_hip_ctx* c, int B, const float* feats, int num_bits, int32_t* idx) {
  CallScope s(c, B);
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < 46; ++k) idx[b * 46 + k] = k < num_bits / 4 ? (((int)feats[b * 64 + k] + k) & 15) : -1;
  return 0;
}
int lyra_hip_rvq_decode(lyra_hip_ctx* c, int B, const int32_t* idx, float* feats) {
  CallScope s(c, B);
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < 64; ++i) feats[b * 64 + i] = (float)(idx[b * 46 + (i % 46)] + 2 * i);
  return 0;
}
}
