// api.hip -- the C ABI of include/lyra_hip.h: context, scratch, launches.  No CPU fallback anywhere:
// every entry point either runs the gfx950 kernels or fails with an error code.
//
// Four HIP streams per context: the ENCODE side (extract, rvq_encode, the extractor of encode), the DECODE side
// (rvq_decode, generate, decode, logmel), one for the quantizer of the `_dev` encode calls (see encq_begin) and one for
// the decoder-side NoiseEstimator of the `_dev` calls (see noise_dev_begin).
// Encoder and decoder state are disjoint, so decode of step i overlaps the extractor of step i+1; every stage kernel
// is a chain of short dependent phases, and two chains in flight fill each other's bubbles.  Ordering: a decode-side
// call waits (on the GPU) for every earlier encode-side call; encode-side outputs never overtake any decode-side call
// but the most recent one (include/lyra_hip.h "Streams").
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/lyra_hip.h"
#include "model.h"
#include "tflite_pack.h"

using namespace lyra;

namespace {
std::string g_create_error;
std::mutex g_create_mu;
}  // namespace

struct lyra_hip_ctx {
  int device = 0;
  int max_streams = 0;
  int mode = 0;
  static constexpr int KMAX = 8;
  int nsub = 1;                    // sub-batches a `_dev` call is split into (independent stream pairs)
  hipStream_t se[KMAX] = {};       // encode side
  hipStream_t sd[KMAX] = {};       // decode side
  hipStream_t sq[KMAX] = {};       // the quantizer of the `_dev` encode calls (see encq_begin)
  hipStream_t sn = nullptr;        // the decoder-side NoiseEstimator of lyra_hip_noise_receive_dev (see noise_dev_begin)
  hipEvent_t ev_noise[2] = {};     // end of the two latest decoder-side `_dev` noise calls on sn, by parity
  long n_noise_calls = 0;          // decoder-side `_dev` noise calls so far
  long noise_done_dec = 0;         // the decode-side streams are ordered after this many of them ...
  long cover_sq[KMAX] = {};        // ... and the latest quantizer record of chunk k after this many (see noise_dev_begin)
  hipEvent_t ev_encs[3][KMAX] = {};// [0], [1]: `_dev` encode calls by parity, recorded on sq[k] after the quantizer;
                                   // [2]: every other encode-side call, recorded on se[k]
  // The decode side has to see BOTH kinds of encode-side work: what ran last on se[k] and the latest quantizer on sq[k]
  // (an extract_dev after an encode_dev does not order the quantizer's packet writes).  Sequence numbers say which of
  // the two a decode-side stream has not waited for yet, so the steady state still costs one wait per call.
  long seq_se[KMAX] = {}, seq_sq[KMAX] = {};                 // records so far on se[k] (ev_encs[2][k]) / sq[k]
  int sq_slot[KMAX] = {};                                     // ev_encs slot (0/1) of the latest quantizer of chunk k
  long seen_se[KMAX + 1][KMAX] = {}, seen_sq[KMAX + 1][KMAX] = {};   // [decode stream (KMAX = sn)][chunk]: number already waited for
  int encq_nk[2] = {1, 1};                                    // chunks of the `_dev` encode call of each parity
  hipEvent_t ev_feat[KMAX] = {};   // features of the current `_dev` encode call ready on se[k]
  long n_encq_calls = 0;           // `_dev` encode calls so far (parity selects the feature buffer and ev_encs slot)
  hipEvent_t ev_dec[2][KMAX] = {}; // end of the two latest decode-side calls on sd[k]
  long n_dec_calls = 0;
  int rvq_wide = 0;                // LYRA_HIP_RVQ_WIDE: the 104 KB / 244-VGPR quantizer kernel (experiment)
  int fused = 0;                   // bit 0: encoder side in one launch, bit 1: decoder side (LYRA_HIP_FUSED)
  bool serial = false;             // lyra_hip_set_serial: encode side also waits for the latest decode-side call
  hipEvent_t ev_caller = nullptr;  // scratch event for lyra_hip_wait_for_stream / lyra_hip_stream_wait
  Model model;
  uint8_t* d_state = nullptr;       // one allocation, carved into per-kernel regions (state_layout.h)
  StateMap sm = {};
  int cw[16] = {};                  // code_warm_bytes per kernel id
  // scratch, sized for `cap` frames
  int cap = 0;
  int32_t* d_ids = nullptr;      // encode-side staging of host ids
  int32_t* d_ids_dec = nullptr;  // decode-side staging of host ids
  int16_t* d_pcm_in = nullptr;
  static constexpr int RS_RING = 3;
  int16_t* d_rs16[RS_RING] = {};  // run_steps: the input resampler's 16 kHz hops, by step mod 3 (resample_in_ahead: two hops ahead)
  hipEvent_t ev_rs_in[RS_RING] = {};   // ... and the end of the launch that filled each
  hipEvent_t ev_ahead_order = nullptr, ev_ahead_last = nullptr;
  // sub-batches (nsub > 1) only: a call that is split differently from the previous call of its side joins that call's
  // streams first (cross_begin / dec_side_begin) -- e.g. an unsplit resample_dev behind a split decode_dev
  int dec_nk_slot[2] = {0, 0};    // chunks of the decode-side call recorded in ev_dec[slot]
  int enc_last_nk = 0;            // chunks of the previous encode-side call
  hipEvent_t ev_se_last[KMAX] = {};  // end of the latest encode-side work on se[k]
  bool se_last_set[KMAX] = {};
  bool rs_sn_pending = false;     // run_steps put an output resampler on the noise stream that sd[0] has not been ordered after yet
  bool ahead_unseen = false;      // something ran ahead on sq[0] that se[0] has not been ordered after yet (wait_ahead)
  float* d_e0 = nullptr;     // [cap][4][128]
  float* d_e1 = nullptr;     // [cap][2][256]
  float* d_feat = nullptr;   // [cap][64]
  float* d_feat2 = nullptr;  // [cap][64]  second feature buffer of the `_dev` encode calls (alternating)
  float* d_codes = nullptr;  // [cap][64]
  int32_t* d_idx = nullptr;  // [cap][46]
  uint8_t* d_pkt = nullptr;  // [cap][23]
  float* d_lossy = nullptr;  // [cap][64]
  float* d_d0 = nullptr;     // [cap][4][128]
  float* d_d1 = nullptr;     // [cap][20][64]
  int16_t* d_pcm_out = nullptr;
  float* d_mel = nullptr;    // [cap][160]  decode side / plugin-level log-mel
  float* d_mel_enc = nullptr;  // [cap][160]  encode side (DTX noise estimator)
  int32_t* d_flag_enc = nullptr;  // [cap] is_noise, encode side
  int32_t* d_flag_dec = nullptr;  // [cap] is_noise, decode side
  int32_t* d_live_ids = nullptr;  // [cap] stream ids with noise hops masked to -1 (DTX)
  int32_t* d_live_ids2 = nullptr; // [cap] second mask buffer of lyra_hip_encode_dtx_dev (alternating with the features)
  int32_t* d_pkt_bytes = nullptr; // [cap]
  int16_t* d_rs_in = nullptr;     // [cap][960] resampler staging (host-pointer entry points)
  int16_t* d_rs_out = nullptr;    // [cap][960]
  unsigned long long cng_seed = 0x4C797261ull;   // comfort-noise phase generator seed (lyra_hip_set_cng_seed)
  // decoder twin (lyra_hip_twin_*, twin_api.inc): by-stream-id hop buffers and the staging arenas of its host arguments
  int16_t* d_twin_gan = nullptr;   // [max_streams][320] conditioned hop of the generative model
  int16_t* d_twin_cng = nullptr;   // [max_streams][320] conditioned hop of the comfort-noise generator
  int16_t* d_twin_noise = nullptr; // [max_streams][320] dense: completed received hops for the noise estimator
  int16_t* d_twin_out = nullptr;   // [max_streams][twin_out_n] the request being assembled (internal rate)
  int16_t* d_twin_ext = nullptr;   // [max_streams][twin_ext_n] ... resampled
  size_t twin_out_cap = 0, twin_ext_cap = 0;   // samples allocated
  int twin_out_n = 0;              // row length of the request in flight (0: none)
  float* d_twin_fade = nullptr;    // [TWIN_FADE_N] cross-fade weights
  int32_t* d_twin_iota = nullptr;  // [max_streams] 0, 1, 2, ...
  uint8_t* h_twin_args = nullptr;  // pinned
  // Small host-buffer calls (the per-object plugin contract: B = 1 per blocking call) skip the copy engine: the kernels read
  // their input from and write their output to this pinned, device-mapped arena directly -- three copy packets and their
  // stream bubbles fewer per call (lyra_amd/plugin_demo --bench).  ZC_MAX streams per call; LYRA_HIP_NO_ZEROCOPY=1 turns it off.
  static constexpr int ZC_MAX = 16;
  static constexpr size_t ZC_IDS = 0, ZC_IN = 256, ZC_OUT = 256 + ZC_MAX * 640, ZC_BYTES = 256 + 2 * ZC_MAX * 640;
  uint8_t* h_zc = nullptr;
  bool zc(int B) const { return h_zc && B <= ZC_MAX; }
  uint8_t* d_twin_args = nullptr;
  size_t twin_args_cap = 0, twin_args_used = 0;
  void* pipe = nullptr;            // PipeState (pipe_api.inc): the two-deep pipelined host-buffer calls, created on first use
  size_t lds_pad[6] = {};          // experiment hook, see lds_pad()
  int tile_div[6] = {1, 1, 1, 1, 1, 1};   // tiles per workgroup of each stage kernel (LYRA_TILE_LOOP), see tile_div()
  bool chunk_local = false;                       // see wait_encode_side
  bool ids_stable = false;                        // inside lyra_hip_run_steps_dev, after its first step: see enc_cross_begin
  unsigned* d_rvq_stats = nullptr;                // rvq_encode_kernel: [0] frame-stages on the exact chain, [1] wavefront-stages (debug_read 5)
  uint32_t cu_pat[4] = {0, 0, 0, 0};              // CU mask pattern of the e / d / q / n streams (0: none), see make_stream_kind
  int enc_noise_rate = 16000;                     // what the DTX encoder's NoiseEstimator::Create is given (lyra_hip_set_encoder_sample_rate)
  int last_B_enc = 0, last_B_dec = 0;
  // optional per-kernel timing with HIP events on the launching stream (bench.py roofline leg)
  unsigned profiling = 0;  // bit i set: bracket launches of kernel i
  int prof_every = 1;      // ... every prof_every-th launch of it (lyra_hip_profile_sample)
  long prof_seen[16] = {};
  struct Span { int kid; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  std::vector<uint32_t> id_stamp;  // duplicate detection for host id lists: id_stamp[id] == id_gen <=> seen in this call
  uint32_t id_gen = 0;
  std::string err;
};

namespace {

int fail(lyra_hip_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  else { std::lock_guard<std::mutex> l(g_create_mu); g_create_error = buf; }
  return code;
}

#define HIPCHK(c, expr)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(c, LYRA_HIP_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <class T>
hipError_t dalloc(T** p, size_t n) { return hipMalloc((void**)p, n * sizeof(T)); }

int pipe_sync(lyra_hip_ctx* c);   // the copy streams of the pipelined host-buffer calls (pipe_api.inc)
int sync_all(lyra_hip_ctx* c) {
  { int rc = pipe_sync(c); if (rc) return rc; }
  for (int k = 0; k < lyra_hip_ctx::KMAX; ++k) {
    if (c->se[k]) HIPCHK(c, hipStreamSynchronize(c->se[k]));
    if (c->sd[k]) HIPCHK(c, hipStreamSynchronize(c->sd[k]));
    if (c->sq[k]) HIPCHK(c, hipStreamSynchronize(c->sq[k]));
  }
  if (c->sn) HIPCHK(c, hipStreamSynchronize(c->sn));
  return 0;
}

void twin_free(lyra_hip_ctx* c);
void pipe_free(lyra_hip_ctx* c);
void free_scratch(lyra_hip_ctx* c) {
  void* ps[] = {c->d_ids, c->d_ids_dec, c->d_pcm_in, c->d_e0, c->d_e1, c->d_feat, c->d_feat2, c->d_codes, c->d_idx, c->d_pkt,
                c->d_lossy, c->d_d0, c->d_d1, c->d_pcm_out, c->d_mel, c->d_mel_enc, c->d_flag_enc, c->d_flag_dec,
                c->d_live_ids, c->d_live_ids2, c->d_pkt_bytes, c->d_rs_in, c->d_rs_out};
  for (void* p : ps)
    if (p) (void)hipFree(p);
  c->d_ids = nullptr; c->d_ids_dec = nullptr; c->d_pcm_in = nullptr; c->d_e0 = nullptr; c->d_e1 = nullptr;
  c->d_feat = nullptr; c->d_feat2 = nullptr; c->d_codes = nullptr; c->d_idx = nullptr; c->d_pkt = nullptr; c->d_lossy = nullptr;
  c->d_d0 = nullptr; c->d_d1 = nullptr; c->d_pcm_out = nullptr; c->d_mel = nullptr; c->d_mel_enc = nullptr;
  c->d_flag_enc = nullptr; c->d_flag_dec = nullptr; c->d_live_ids = nullptr; c->d_live_ids2 = nullptr; c->d_pkt_bytes = nullptr;
  c->d_rs_in = nullptr; c->d_rs_out = nullptr;
  for (auto& p : c->d_rs16) { if (p) (void)hipFree(p); p = nullptr; }
  c->cap = 0;
}

int ensure_scratch(lyra_hip_ctx* c, int B) {
  if (B <= c->cap) return 0;
  int rc = sync_all(c);
  if (rc) return rc;
  free_scratch(c);
  size_t n = (size_t)B;
  HIPCHK(c, dalloc(&c->d_ids, n));
  HIPCHK(c, dalloc(&c->d_ids_dec, n));
  HIPCHK(c, dalloc(&c->d_pcm_in, n * 320));
  HIPCHK(c, dalloc(&c->d_e0, n * 4 * 128));
  HIPCHK(c, dalloc(&c->d_e1, n * 2 * 256));
  HIPCHK(c, dalloc(&c->d_feat, n * 64));
  HIPCHK(c, dalloc(&c->d_feat2, n * 64));
  HIPCHK(c, dalloc(&c->d_codes, n * 64));
  HIPCHK(c, dalloc(&c->d_idx, n * 46));
  HIPCHK(c, dalloc(&c->d_pkt, n * 23));
  HIPCHK(c, dalloc(&c->d_lossy, n * 64));
  HIPCHK(c, dalloc(&c->d_d0, n * 4 * 128));
  HIPCHK(c, dalloc(&c->d_d1, n * 20 * 64));
  HIPCHK(c, dalloc(&c->d_pcm_out, n * 320));
  HIPCHK(c, dalloc(&c->d_mel, n * 160));
  HIPCHK(c, dalloc(&c->d_mel_enc, n * 160));
  HIPCHK(c, dalloc(&c->d_flag_enc, n));
  HIPCHK(c, dalloc(&c->d_flag_dec, n));
  HIPCHK(c, dalloc(&c->d_live_ids, n));
  HIPCHK(c, dalloc(&c->d_live_ids2, n));
  HIPCHK(c, dalloc(&c->d_pkt_bytes, n));
  HIPCHK(c, dalloc(&c->d_rs_in, n * 960));
  HIPCHK(c, dalloc(&c->d_rs_out, n * 960));
  for (auto& p : c->d_rs16) HIPCHK(c, dalloc(&p, n * 320));
  c->cap = B;
  return 0;
}

int check_batch(lyra_hip_ctx* c, int B) {
  if (!c) return LYRA_HIP_EINVAL;
  if (B <= 0 || B > c->max_streams) return fail(c, LYRA_HIP_EINVAL, "batch %d outside 1..max_streams (%d)", B, c->max_streams);
  return 0;
}

int check_bits(lyra_hip_ctx* c, int num_bits) {
  // residual_vector_quantizer.cc:79-89,116-126
  if (num_bits > 4 * LYRA_HIP_MAX_STAGES)
    return fail(c, LYRA_HIP_EINVAL, "The number of bits cannot exceed maximum (%d).", 4 * LYRA_HIP_MAX_STAGES);
  if (num_bits <= 0 || num_bits % 4 != 0)
    return fail(c, LYRA_HIP_EINVAL, "The number of bits (%d) has to be divisible by the number of bits per quantizer (4).", num_bits);
  return 0;
}

int check_ids_host(lyra_hip_ctx* c, const int32_t* ids, int B) {
  if (!ids) return fail(c, LYRA_HIP_EINVAL, "stream_ids is null");
  if (c->id_stamp.size() != (size_t)c->max_streams) c->id_stamp.assign(c->max_streams, 0);
  if (++c->id_gen == 0) { std::fill(c->id_stamp.begin(), c->id_stamp.end(), 0u); c->id_gen = 1; }
  for (int i = 0; i < B; ++i) {
    if (ids[i] < 0 || ids[i] >= c->max_streams)
      return fail(c, LYRA_HIP_EINVAL, "stream id %d at position %d outside 0..%d", ids[i], i, c->max_streams - 1);
    if (c->id_stamp[ids[i]] == c->id_gen)   // two frames of one stream in a batch would race on its state
      return fail(c, LYRA_HIP_EINVAL, "stream id %d appears twice in the batch (position %d)", ids[i], i);
    c->id_stamp[ids[i]] = c->id_gen;
  }
  return 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A stream of kind e / d / q / n (0..3): with the kind's CU mask when one is configured (no priority then: the masked
// streams do not compete for CUs), else with the given priority.
hipError_t make_stream_kind(lyra_hip_ctx* c, hipStream_t* s, int kind, int priority) {
  if (!c->cu_pat[kind]) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu <= 0) ncu = 256;
  std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, c->cu_pat[kind]);
  if (hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()) == hipSuccess) return hipSuccess;
  (void)hipGetLastError();   // a runtime / partition mode without CU masking: an ordinary stream (placement is then left to the dispatcher)
  c->cu_pat[kind] = 0;
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority);
}

// EXPERIMENT hook (occupancy): LYRA_HIP_LDS_PAD_<kernel>=bytes asks for that much extra dynamic LDS per workgroup of a
// stage kernel, i.e. fewer of its workgroups per CU and more room for the kernel running beside it.
inline size_t lds_pad(const char* kernel) {
  char name[64];
  snprintf(name, sizeof name, "LYRA_HIP_LDS_PAD_%s", kernel);
  const char* v = getenv(name);
  return v ? (size_t)atol(v) : 0;
}

// LYRA_HIP_TILE_DIV_<kernel>=k launches a stage kernel as k back-to-back slices of 1/k of its tiles each (tile0 = first
// tile of the slice): at any time the kernel holds at most 1/k of the CUs' LDS and wave slots, the rest stays free for
// the other chain's kernel.
inline int tile_div(const char* kernel) {
  char name[64];
  snprintf(name, sizeof name, "LYRA_HIP_TILE_DIV_%s", kernel);
  const char* v = getenv(name);
  const int k = v ? atoi(v) : 1;
  return k < 1 ? 1 : k;
}

// machine-code size of each kernel (generated at build time by code_sizes.sh from the kernel objects)
struct CodeSize { const char* name; int bytes; int getpc; };   // getpc: offset of the kernel's s_getpc_b64 (= bytes: none found)
const CodeSize kCodeSizes[] = {
#include "code_sizes.inc"
    {"", 0, 0}};
// What a kernel is told to warm: from the 128-byte line of its s_getpc to the end of the function.  The offset of that
// instruction is read back from the kernel object at build time (code_sizes.sh), so the range ends inside the function
// wherever the compiler scheduled the s_getpc; disabled with LYRA_HIP_NO_CODE_WARM=1
int code_warm_bytes(const char* kernel) {
  static const bool off = getenv("LYRA_HIP_NO_CODE_WARM") != nullptr;
  if (off) return 0;
  for (const CodeSize& c : kCodeSizes)
    if (strcmp(c.name, kernel) == 0) return c.bytes > 2048 && c.bytes > c.getpc ? c.bytes - c.getpc : 0;
  return 0;
}

enum { K_ENC_S0, K_ENC_S1, K_ENC_S2, K_RVQ_ENC, K_RVQ_DEC, K_DEC_S0, K_DEC_S1, K_DEC_S2, K_LOGMEL, K_NOISE, K_RESAMPLE,
       K_CNG, K_ENC_SIDE, K_DEC_SIDE, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"enc_s0_kernel", "enc_s1_kernel", "enc_s2_kernel", "rvq_encode_kernel",
                                           "rvq_decode_kernel", "dec_s0_kernel", "dec_s1_kernel", "dec_s2_kernel",
                                           "logmel_kernel", "logmel_noise_kernel", "resample_kernel", "cng_kernel",
                                           "enc_side_kernel", "dec_side_kernel"};

hipEvent_t take_event(lyra_hip_ctx* c) {
  if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
// Timing only: a failed record here loses one sample (the span is dropped), never an ordering edge.
struct ProfScope {
  lyra_hip_ctx* c; int kid; hipStream_t s; hipEvent_t a = nullptr;
  ProfScope(lyra_hip_ctx* c_, int kid_, hipStream_t s_) : c(c_), kid(kid_), s(s_) {
    if ((c->profiling & (1u << kid)) && (c->prof_seen[kid]++ % c->prof_every) == 0) {
      a = take_event(c);
      if (a && hipEventRecord(a, s) != hipSuccess) { c->event_pool.push_back(a); a = nullptr; }
    }
  }
  ~ProfScope() {
    if (!a) return;
    hipEvent_t b = take_event(c);
    if (b && hipEventRecord(b, s) == hipSuccess) { c->spans.push_back({kid, a, b}); return; }
    c->event_pool.push_back(a);
    if (b) c->event_pool.push_back(b);
  }
};

// ---- launches -----------------------------------------------------------------------------------------
// Ordering between the two sides (see include/lyra_hip.h "Streams"):
//  * decode-side work on chunk k waits for all encode-side work enqueued so far;
//  * encode-side work on chunk k waits for every decode-side call except the most recent one, so a caller that
//    alternates two buffers never has a buffer rewritten while a pending decode still reads it.
// These edges ARE the encode -> decode dependency: every record / wait is checked, a failure fails the call.
// Chunk k of a split call (LYRA_HIP_SUBBATCHES > 1) runs on a stream of its own.  Its streams' state was last written by
// whichever chunk of the PREVIOUS call on this side held those streams: the same chunk only if the caller lists the same ids
// in the same order again.  That is known inside lyra_hip_run_steps_dev (one id list for all its steps: ids_stable from the
// second step on) -- there the chunks stay independent pipelines; every other call waits for all chunks of the call before it.
// (Round 6: with a stream order that changed from call to call, chunk 0 of call n + 1 overtook chunk 1 of call n on the
// streams that had moved between them -- tests/test_gpu_pipelined.py, found under a slower kernel build.)
int enc_cross_begin(lyra_hip_ctx* c, int k, int nk_now) {
  if (c->nsub > 1 && c->enc_last_nk && (c->enc_last_nk != nk_now || !c->ids_stable))
    for (int j = 0; j < c->nsub; ++j)
      if (c->se_last_set[j] && j != k) HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_se_last[j], 0));
  return 0;
}
int enc_cross_done(lyra_hip_ctx* c, int k) {
  if (c->nsub > 1) {
    HIPCHK(c, hipEventRecord(c->ev_se_last[k], c->se[k]));
    c->se_last_set[k] = true;
  }
  return 0;
}
int enc_side_begin(lyra_hip_ctx* c, int k, int nk_now = 1) {
  int rc = enc_cross_begin(c, k, nk_now);
  if (rc) return rc;
  if (c->n_dec_calls >= 2) HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_dec[c->n_dec_calls & 1][k], 0));
  if (c->serial && c->n_dec_calls >= 1)
    HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_dec[(c->n_dec_calls - 1) & 1][k], 0));
  return 0;
}
int enc_side_done(lyra_hip_ctx* c, int k, int nk = 1) {   // nk > 1: the caller sets enc_last_nk after its last chunk
  HIPCHK(c, hipEventRecord(c->ev_encs[2][k], c->se[k]));
  c->seq_se[k]++;
  if (nk == 1) c->enc_last_nk = 1;
  return enc_cross_done(c, k);
}
// The `_dev` encode calls (lyra_hip_encode_dev, lyra_hip_encode_dtx_dev) run the feature extractor on se[k] and the
// quantizer on a third stream sq[k]: rvq_encode is a 46-stage dependent chain on one wavefront per SIMD that leaves
// the chip nearly idle.  Behind the extractor on se[k] it kept the next call's extractor waiting (rocprofv3 timeline:
// ~50 us of quantizer + ~12 us of cross-stream event latency per step with nothing else running); in front of the
// decoder stages on sd[k] it lengthened the chain that paces the pipeline.  On its own stream it runs underneath the
// next call's extractor and the previous call's decoder stages, which then are two chains of equal length that never
// wait for each other (B = 4096: 0.338 -> 0.315 ms per step).  The features travel through two alternating buffers.
int encq_begin(lyra_hip_ctx* c, int k, int nk_now = 1) {
  int rc = enc_cross_begin(c, k, nk_now);
  if (rc) return rc;
  if (c->serial && c->n_dec_calls >= 1)
    HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_dec[(c->n_dec_calls - 1) & 1][k], 0));
  return 0;
}
// The feature buffer of this call was last read by the quantizer of the call before the previous one: the extractor's
// last stage (the only writer) waits for it, stages 0 and 1 do not.  That call may have been split differently
// (LYRA_HIP_SUBBATCHES > 1 and a batch below the split threshold, or the unsplit DTX path): then chunk k of this call
// overlaps several of its chunks and waits for all of them.
struct EventList { hipEvent_t e[lyra_hip_ctx::KMAX]; int n = 0; };
EventList encq_buffer_free(lyra_hip_ctx* c, int k, int nk_now) {
  EventList l;
#ifdef LYRA_ABL_NO_FEAT_WAIT   // TIMING-ONLY ablation (a race): the extractor's last stage does not wait -- what a deeper feature ring
  return l;                    // with one wait per several steps could return at most (profiles/r06_ab_t1_featwait.txt)
#endif
  if (c->n_encq_calls < 2) return l;
  const int p = (int)(c->n_encq_calls & 1), nk_then = c->encq_nk[p];
  if (nk_then == nk_now) { l.e[l.n++] = c->ev_encs[p][k]; return l; }
  for (int j = 0; j < nk_then; ++j) l.e[l.n++] = c->ev_encs[p][j];
  return l;
}
float* encq_features(lyra_hip_ctx* c) { return (c->n_encq_calls & 1) ? c->d_feat2 : c->d_feat; }
int encq_handoff(lyra_hip_ctx* c, int k) {   // extractor done on se[k] -> quantizer may start on sq[k]
  HIPCHK(c, hipEventRecord(c->ev_feat[k], c->se[k]));
  { int rc = enc_cross_done(c, k); if (rc) return rc; }
  HIPCHK(c, hipStreamWaitEvent(c->sq[k], c->ev_feat[k], 0));
  // the packets: the two-buffer rule of include/lyra_hip.h "Streams" (2), as enc_side_begin does it for se[k]
  if (c->n_dec_calls >= 2) HIPCHK(c, hipStreamWaitEvent(c->sq[k], c->ev_dec[c->n_dec_calls & 1][k], 0));
  if (c->serial && c->n_dec_calls >= 1)
    HIPCHK(c, hipStreamWaitEvent(c->sq[k], c->ev_dec[(c->n_dec_calls - 1) & 1][k], 0));
  if (c->n_noise_calls - 1 > std::max(c->cover_sq[k], c->noise_done_dec)) {   // see noise_dev_begin
    HIPCHK(c, hipStreamWaitEvent(c->sq[k], c->ev_noise[(c->n_noise_calls - 2) & 1], 0));
    c->cover_sq[k] = c->n_noise_calls - 1;
  }
  return 0;
}
int encq_done(lyra_hip_ctx* c, int k) {
  const int p = (int)(c->n_encq_calls & 1);
  HIPCHK(c, hipEventRecord(c->ev_encs[p][k], c->sq[k]));
  c->sq_slot[k] = p;
  c->seq_sq[k]++;
  return 0;
}
// include/lyra_hip.h "Streams" (1): ordered after EVERY earlier encode-side call -- the latest record on se[j] and the
// latest quantizer on sq[j] of every chunk (earlier records of a stream are implied by its latest one).
// `who`: index of the waiting stream in seen_* (decode stream k, or KMAX for the noise stream).
int wait_encode_side(lyra_hip_ctx* c, hipStream_t s, int who) {
  for (int j = 0; j < c->nsub; ++j) {
    // inside lyra_hip_run_steps_dev every call of a step is split the same way over the same stream ids and buffers:
    // decode chunk k reads what encode chunk k wrote and nothing else, so the sub-batches are independent pipelines
    if (c->chunk_local && who < lyra_hip_ctx::KMAX && j != who) continue;
    if (c->seen_se[who][j] != c->seq_se[j]) {
      HIPCHK(c, hipStreamWaitEvent(s, c->ev_encs[2][j], 0));
      c->seen_se[who][j] = c->seq_se[j];
    }
    if (c->seen_sq[who][j] != c->seq_sq[j]) {
      HIPCHK(c, hipStreamWaitEvent(s, c->ev_encs[c->sq_slot[j]][j], 0));
      c->seen_sq[who][j] = c->seq_sq[j];
    }
  }
  return 0;
}
// The decoder-side NoiseEstimator of the `_dev` path (lyra_decoder.cc:304-311 feeds it every decoded hop) runs on a
// stream of its own, behind the decoder's last stage, underneath the next step -- on the decoder's stream it would
// lengthen the chain that paces the pipeline (round 2: +68 us per step for a 33 us kernel).  What it reads is the
// decoder's PCM buffer, which by the two-buffer rule is rewritten by the decode-side call after the next one: that
// call has to be ordered after this kernel.  In the encode+decode pipeline the edge rides on the quantizer stream,
// which has slack: the quantizer of a `_dev` encode call waits for all noise calls but the most recent one
// (encq_handoff), and the decoder stages wait for that quantizer anyway; a decode-side call that is not covered that way
// (decode-only loops, generate_dev) waits itself.  noise_done_dec / cover_sq[] do the bookkeeping.
int dec_side_begin(lyra_hip_ctx* c, int k, int nk_now = 1) {
  int rc = wait_encode_side(c, c->sd[k], k);
  if (rc) return rc;
  if (c->nsub > 1 && c->n_dec_calls >= 1) {   // split differently from the previous decode-side call: after all of it
    const int prev = (int)((c->n_dec_calls - 1) & 1);
    if (c->dec_nk_slot[prev] != nk_now || !c->ids_stable)   // (or the ids may have moved between chunks: enc_cross_begin)
      for (int j = 0; j < c->nsub; ++j) HIPCHK(c, hipStreamWaitEvent(c->sd[k], c->ev_dec[prev][j], 0));
  }
  const long required = c->n_noise_calls - 1;   // all decoder-side noise calls but the most recent one
  if (required > c->noise_done_dec) {
    long covered = 0;
    for (int j = 0; j < c->nsub; ++j) covered = std::max(covered, c->cover_sq[j]);
    if (covered < required) {
      for (int kk = 0; kk < c->nsub; ++kk)
        HIPCHK(c, hipStreamWaitEvent(c->sd[kk], c->ev_noise[(required - 1) & 1], 0));
      covered = required;
    }
    // (covered through sq: every decode stream has waited for the latest quantizer record of every chunk above)
    c->noise_done_dec = covered;
  }
  return 0;
}
int noise_dev_begin(lyra_hip_ctx* c) {   // sn: after the latest decode-side call of every chunk, and rule (1)
  int rc = wait_encode_side(c, c->sn, lyra_hip_ctx::KMAX);
  if (rc) return rc;
  if (c->n_dec_calls >= 1)
    for (int j = 0; j < c->nsub; ++j) HIPCHK(c, hipStreamWaitEvent(c->sn, c->ev_dec[(c->n_dec_calls - 1) & 1][j], 0));
  return 0;
}
int noise_dev_done(lyra_hip_ctx* c) {
  HIPCHK(c, hipEventRecord(c->ev_noise[c->n_noise_calls & 1], c->sn));
  c->n_noise_calls++;
  return 0;
}
int dec_side_done(lyra_hip_ctx* c, int k, int nk = 0) {
  const int slot = (int)(c->n_dec_calls & 1);
  c->dec_nk_slot[slot] = nk;
  if (nk == 1) {  // an unsplit call stands for every chunk
    for (int j = 0; j < c->nsub; ++j) HIPCHK(c, hipEventRecord(c->ev_dec[slot][j], c->sd[0]));
  } else {
    HIPCHK(c, hipEventRecord(c->ev_dec[slot][k], c->sd[k]));
  }
  return 0;
}
// `_dev` entry points run on the context's device whatever the caller's current device is (a process may hold
// contexts on several GPUs); the caller's device is restored on return.
struct DeviceScope {
  int prev = -1; bool ok = true;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define DEVSCOPE(c)                   \
  DeviceScope devscope_((c)->device); \
  if (!devscope_.ok) return fail(c, LYRA_HIP_EHIP, "hipSetDevice(%d) failed", (c)->device)

// chunk k of a batch of B frames: [lo, lo + n), multiples of 16 streams so tiles stay whole
void chunk_of(const lyra_hip_ctx* c, int B, int k, int* lo, int* n) {
  int per = (B / c->nsub + 15) / 16 * 16;
  int a = per * k, b = (k == c->nsub - 1) ? B : per * (k + 1);
  if (a > B) a = B;
  if (b > B) b = B;
  *lo = a;
  *n = b - a;
}
int chunks_for(const lyra_hip_ctx* c, int B) { return (c->nsub > 1 && B >= 64 * c->nsub) ? c->nsub : 1; }

// before_s2: an event the LAST stage (the only one that writes d_feat) has to wait for -- the earlier stages run ahead.
int launch_extract(lyra_hip_ctx* c, int k, int lo, const int32_t* d_ids, int B, const int16_t* d_pcm, float* d_feat,
                   const EventList& before_s2 = EventList()) {
  const Model& M = c->model;
  hipStream_t st_ = c->se[k];
  float* e0 = c->d_e0 + (size_t)lo * 512;
  float* e1 = c->d_e1 + (size_t)lo * 512;
  float* codes = c->d_codes + (size_t)lo * 64;
#ifdef LYRA_PARKED
  if (c->fused & 1) {   // the whole side in one launch (enc_side_kernel.hip)
    for (int i = 0; i < before_s2.n; ++i) HIPCHK(c, hipStreamWaitEvent(st_, before_s2.e[i], 0));
    { ProfScope ps(c, K_ENC_SIDE, st_);
      hipLaunchKernelGGL(c->mode == 2 ? enc_side_xn_kernel : c->mode ? enc_side_dr_kernel : enc_side_kernel, dim3(cdiv(B, 8)), dim3(512), enc_side_lds_bytes(), st_,
                         M.d_enc0, M.d_enc1, M.d_enc2, d_pcm, d_ids, B, c->sm.base[st::R_E0], c->sm.base[st::R_E1],
                         c->sm.base[st::R_E2], e0, e1, d_feat, codes, c->cw[K_ENC_SIDE]); }
    HIPCHK(c, hipGetLastError());
    c->last_B_enc = B;
    return 0;
  }
#endif
  { ProfScope ps(c, K_ENC_S0, st_);
    for (int nt = cdiv(B, enc_s0_streams_per_wg()), g = cdiv(nt, c->tile_div[0]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(enc_s0_kernel, dim3(std::min(g, nt - t0)), dim3(enc_s0_threads()), enc_s0_lds_bytes() + c->lds_pad[0], st_,
                         M.d_enc0, d_pcm, d_ids, B, c->sm.base[st::R_E0], e0, c->cw[K_ENC_S0], t0); }
#ifdef LYRA_PARKED
  if ((c->fused & 4) && c->mode == 2) {   // stages 1 + 2 in one launch
    for (int i = 0; i < before_s2.n; ++i) HIPCHK(c, hipStreamWaitEvent(st_, before_s2.e[i], 0));
    { ProfScope ps(c, K_ENC_S1, st_);
      hipLaunchKernelGGL(enc_s12_xn_kernel, dim3(cdiv(B, 8)), dim3(512), enc_s12_lds_bytes(), st_, M.d_enc1, M.d_enc2, e0, d_ids, B,
                         c->sm.base[st::R_E1], c->sm.base[st::R_E2], e1, d_feat, codes, c->cw[K_ENC_S1]); }
    HIPCHK(c, hipGetLastError());
    c->last_B_enc = B;
    return 0;
  }
#endif
  { ProfScope ps(c, K_ENC_S1, st_);
    for (int nt = cdiv(B, enc_s1_streams_per_wg()), g = cdiv(nt, c->tile_div[1]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(enc_s1_kernel, dim3(std::min(g, nt - t0)), dim3(enc_s1_threads()), enc_s1_lds_bytes() + c->lds_pad[1], st_,
                         M.d_enc1, e0, d_ids, B, c->sm.base[st::R_E1], e1, c->cw[K_ENC_S1], t0); }
  for (int i = 0; i < before_s2.n; ++i) HIPCHK(c, hipStreamWaitEvent(st_, before_s2.e[i], 0));
  { ProfScope ps(c, K_ENC_S2, st_);
    for (int nt = cdiv(B, enc_s2_streams_per_wg()), g = cdiv(nt, c->tile_div[2]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(c->mode == 2 ? enc_s2_xn_kernel : c->mode == 3 ? enc_s2_bm_kernel : c->mode ? enc_s2_dr_kernel : enc_s2_kernel, dim3(std::min(g, nt - t0)), dim3(512),
                         enc_s2_lds_bytes() + c->lds_pad[2], st_, M.d_enc2, e1, d_ids, B, c->sm.base[st::R_E2], d_feat, codes,
                         c->cw[K_ENC_S2], t0); }
  HIPCHK(c, hipGetLastError());
  c->last_B_enc = B;
  return 0;
}

int launch_rvq_encode(lyra_hip_ctx* c, int k, int B, const float* d_feat, int num_stages, int32_t* d_idx,
                      uint8_t* d_pkt, const int32_t* d_mask_ids = nullptr, int32_t* d_pkt_bytes = nullptr,
                      bool on_quantizer_stream = false) {
  hipStream_t st_ = on_quantizer_stream ? c->sq[k] : c->se[k];
  { ProfScope ps(c, K_RVQ_ENC, st_);
#ifdef LYRA_PARKED
    if (c->rvq_wide) {   // 1: the all-exact chain kernel of rounds 2-3, 2: its 104 KB / 244-VGPR form
      hipLaunchKernelGGL(c->rvq_wide == 2 ? rvq_encode_wide_kernel : rvq_encode_chain_kernel, dim3(cdiv(B, 16)), dim3(256), 0, st_,
                         c->model.cb, d_feat, B, num_stages, d_idx, d_pkt, d_mask_ids, d_pkt_bytes);
    } else
#endif
    hipLaunchKernelGGL(rvq_encode_kernel, dim3(cdiv(B, 16)), dim3(64), 0, st_, c->model.cb, c->model.cbn, d_feat, B,
                       num_stages, d_idx, d_pkt, d_mask_ids, d_pkt_bytes, c->d_rvq_stats); }
  HIPCHK(c, hipGetLastError());
  return 0;
}

int launch_rvq_decode(lyra_hip_ctx* c, int k, int B, const int32_t* d_idx, const uint8_t* d_pkt, int num_stages,
                      float* d_feat) {
  { ProfScope ps(c, K_RVQ_DEC, c->sd[k]);
    hipLaunchKernelGGL(rvq_decode_kernel, dim3(cdiv(B, 4)), dim3(256), 0, c->sd[k], c->model.cb, d_idx, d_pkt,
                       num_stages, B, d_feat); }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// d_feat != nullptr: features -> PCM.  d_feat == nullptr: packets -> PCM with the RVQ decode fused into stage 0.
int launch_generate(lyra_hip_ctx* c, int k, int lo, const int32_t* d_ids, int B, const float* d_feat, int16_t* d_pcm,
                    const uint8_t* d_pkt = nullptr, int num_stages = 0) {
  const Model& M = c->model;
  hipStream_t st_ = c->sd[k];
  float* d0 = c->d_d0 + (size_t)lo * 512;
  float* d1 = c->d_d1 + (size_t)lo * 1280;
#ifdef LYRA_PARKED
  if (c->fused & 2) {   // the whole side in one launch (dec_side_kernel.hip)
    { ProfScope ps(c, K_DEC_SIDE, st_);
      hipLaunchKernelGGL(c->mode == 2 ? dec_side_xn_kernel : c->mode ? dec_side_dr_kernel : dec_side_kernel, dim3(cdiv(B, 8)), dim3(512), dec_side_lds_bytes(), st_,
                         M.d_dec0, M.d_dec1, M.d_dec2, d_feat, d_ids, B, c->sm.base[st::R_D0], c->sm.base[st::R_D1],
                         c->sm.base[st::R_D2], d0, d1, d_pcm, d_pkt, num_stages, M.cb, c->cw[K_DEC_SIDE]); }
    HIPCHK(c, hipGetLastError());
    c->last_B_dec = B;
    return 0;
  }
#endif
#ifdef LYRA_PARKED
  if ((c->fused & 8) && c->mode == 2) {   // stages 0 + 1 in one launch
    { ProfScope ps(c, K_DEC_S0, st_);
      hipLaunchKernelGGL(dec_s01_xn_kernel, dim3(cdiv(B, 8)), dim3(512), dec_s01_lds_bytes(), st_, M.d_dec0, M.d_dec1, d_feat, d_ids, B,
                         c->sm.base[st::R_D0], c->sm.base[st::R_D1], d0, d1, d_pkt, num_stages, M.cb, c->cw[K_DEC_S0]); }
    { ProfScope ps(c, K_DEC_S2, st_);
      hipLaunchKernelGGL(dec_s2_kernel, dim3(cdiv(B, dec_s2_streams_per_wg())), dim3(dec_s2_threads()), dec_s2_lds_bytes() + c->lds_pad[5], st_,
                         M.d_dec2, d1, d_ids, B, c->sm.base[st::R_D2], d_pcm, c->cw[K_DEC_S2], 0); }
    HIPCHK(c, hipGetLastError());
    c->last_B_dec = B;
    return 0;
  }
#endif
  { ProfScope ps(c, K_DEC_S0, st_);
    for (int nt = cdiv(B, dec_s0_streams_per_wg()), g = cdiv(nt, c->tile_div[3]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(c->mode == 2 ? dec_s0_xn_kernel : c->mode == 3 ? dec_s0_bm_kernel : c->mode ? dec_s0_dr_kernel : dec_s0_kernel, dim3(std::min(g, nt - t0)), dim3(512),
                         dec_s0_lds_bytes() + c->lds_pad[3], st_,
                         M.d_dec0, d_feat, d_ids, B, c->sm.base[st::R_D0], d0, d_pkt, num_stages, M.cb,
                         c->cw[K_DEC_S0], t0); }
  { ProfScope ps(c, K_DEC_S1, st_);
    for (int nt = cdiv(B, dec_s1_streams_per_wg()), g = cdiv(nt, c->tile_div[4]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(dec_s1_kernel, dim3(std::min(g, nt - t0)), dim3(dec_s1_threads()), dec_s1_lds_bytes() + c->lds_pad[4], st_,
                         M.d_dec1, d0, d_ids, B, c->sm.base[st::R_D1], d1, c->cw[K_DEC_S1], t0); }
  { ProfScope ps(c, K_DEC_S2, st_);
    for (int nt = cdiv(B, dec_s2_streams_per_wg()), g = cdiv(nt, c->tile_div[5]), t0 = 0; t0 < nt; t0 += g)
      hipLaunchKernelGGL(dec_s2_kernel, dim3(std::min(g, nt - t0)), dim3(dec_s2_threads()), dec_s2_lds_bytes() + c->lds_pad[5], st_,
                         M.d_dec2, d1, d_ids, B, c->sm.base[st::R_D2], d_pcm, c->cw[K_DEC_S2], t0); }
  HIPCHK(c, hipGetLastError());
  c->last_B_dec = B;
  return 0;
}

// NoiseEstimator::Create's constants (noise_estimator.cc:96-124).  The hop duration is derived from the sample rate the
// CALLER passes: 16 kHz by the decoder (lyra_decoder.cc:129-131), the EXTERNAL rate -- with the internal hop of 320
// samples -- by a DTX encoder (lyra_encoder.cc:82-85), so an 8 / 32 / 48 kHz encoder updates its noise estimate every
// 25 / 100 / 150 hops and decays its bounds accordingly.  (Found by running the reference's own classes, oracle/_ref.)
NoiseP noise_params(int sample_rate_hz = 16000) {
  const float secs_per_hop = 320.f / sample_rate_hz;
  NoiseP p;
  p.hops_per_update = (int)roundf(1.f / secs_per_hop);
  p.max_smoothing = powf(0.5f, secs_per_hop / 0.7f);
  p.bound_decay = powf(0.5f, secs_per_hop / 1.f);
  return p;
}

int launch_logmel(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm, float* d_mel) {
  { ProfScope ps(c, K_LOGMEL, c->sd[0]);
    hipLaunchKernelGGL(logmel_kernel, dim3(cdiv(B, 2)), dim3(256), logmel_lds_bytes(), c->sd[0], c->model.d_mel, d_pcm,
                       d_ids, B, c->sm.base[st::R_MEL], (int)st::MEL_BYTES, (int)st::M_PREV, d_mel, 0, noise_params(),
                       (int32_t*)nullptr, (int32_t*)nullptr); }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// NoiseEstimator::ReceiveSamples for one full hop of B streams (noise_estimator.cc:144-173) in ONE launch: the
// estimator's own log-mel front end with the decision + recurrence as its tail (logmel_kernel, noise_tail = 1).
// side 0 = encoder (DTX), 1 = decoder; the caller picks the stream.
int launch_noise(lyra_hip_ctx* c, int side, hipStream_t st_, const int32_t* d_ids, int B, const int16_t* d_pcm,
                 int32_t* d_is_noise, int32_t* d_masked_ids) {
  uint8_t* region = c->sm.base[side == 0 ? st::R_NOISE_E : st::R_NOISE_D];
  // the estimator's extractor is created with the rate NoiseEstimator::Create is given (noise_estimator.cc:104-106): its
  // mel filterbank follows the DTX encoder's external rate, like the time constants
  const int rate = side == 0 ? c->enc_noise_rate : 16000;
  const MelP* melp = c->model.d_mel_rate[rate == 8000 ? 0 : rate == 32000 ? 2 : rate == 48000 ? 3 : 1];
  { ProfScope ps(c, K_NOISE, st_);
    static const size_t pad = lds_pad("logmel_noise");   // experiment hook (fewer estimator workgroups per CU)
    hipLaunchKernelGGL(logmel_kernel, dim3(cdiv(B, 2)), dim3(256), logmel_lds_bytes() + pad, st_, melp, d_pcm, d_ids,
                       B, region, (int)st::NOISE_BYTES, (int)st::N_PREV, (float*)nullptr, 1,
                       noise_params(rate), d_is_noise, d_masked_ids); }
  HIPCHK(c, hipGetLastError());
  return 0;
}

template <class K>
hipError_t set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

}  // namespace

extern "C" {

static int create_impl(const char* model_dir, const void* image, size_t image_bytes, int device, int max_streams,
                       int requant_mode, lyra_hip_ctx** out);

static int create_guarded(const char* model_dir, const void* image, size_t image_bytes, int device, int max_streams,
                          int requant_mode, lyra_hip_ctx** out) {
  try {
    return create_impl(model_dir, image, image_bytes, device, max_streams, requant_mode, out);
  } catch (const std::bad_alloc&) {      // no exception may cross the C boundary
    if (out) *out = nullptr;
    return fail(nullptr, LYRA_HIP_ENOMEM, "out of host memory while loading the model");
  } catch (const std::exception& e) {
    if (out) *out = nullptr;
    return fail(nullptr, LYRA_HIP_EMODEL, "model load failed: %s", e.what());
  }
}

int lyra_hip_create(const char* model_dir, int device, int max_streams, int requant_mode, lyra_hip_ctx** out) {
  if (!model_dir) return fail(nullptr, LYRA_HIP_EINVAL, "lyra_hip_create: bad argument");
  return create_guarded(model_dir, nullptr, 0, device, max_streams, requant_mode, out);
}

int lyra_hip_create_from_image(const void* image, size_t image_bytes, int device, int max_streams, int requant_mode,
                               lyra_hip_ctx** out) {
  if (!image || image_bytes == 0) return fail(nullptr, LYRA_HIP_EINVAL, "lyra_hip_create_from_image: bad argument");
  return create_guarded(nullptr, image, image_bytes, device, max_streams, requant_mode, out);
}

static int create_impl(const char* model_dir, const void* image, size_t image_bytes, int device, int max_streams,
                       int requant_mode, lyra_hip_ctx** out) {
  if (!out || max_streams <= 0 || (requant_mode < 0 || requant_mode > 3))
    return fail(nullptr, LYRA_HIP_EINVAL, "lyra_hip_create: bad argument");
  *out = nullptr;
  {   // the kernels address a stream's state as a 32-bit byte offset into its region (lyra_dev.h goff): id * region bytes
    int worst = 0;
    for (int r = 0; r < st::R_COUNT; ++r) worst = std::max(worst, st::REGION_BYTES[r]);
    const long long cap = (1ll << 32) / worst;
    if (max_streams > cap)
      return fail(nullptr, LYRA_HIP_EINVAL, "lyra_hip_create: max_streams %d, at most %lld streams per context (32-bit byte "
                  "offsets into the per-stream state: use several contexts)", max_streams, cap);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(nullptr, LYRA_HIP_ENODEV, "no HIP device %d (count %d); this library has no CPU path", device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, LYRA_HIP_ENODEV, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, LYRA_HIP_ENODEV, "device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, LYRA_HIP_EHIP, "hipSetDevice failed");

  Pack pk;
  std::string err;
  // Either the pre-packed container or -- as the reference's factories get it -- a model directory with the three
  // .tflite graphs and lyra_config.binarypb (lyra_components.cc:42-55, lyra_config.cc:55-58), converted in memory.
  if (image) {   // an in-memory LYRAPK01 image (e.g. received over an RCCL broadcast)
    const uint8_t* p = static_cast<const uint8_t*>(image);
    if (!pk.adopt(std::vector<uint8_t>(p, p + image_bytes), &err)) return fail(nullptr, LYRA_HIP_EMODEL, "%s", err.c_str());
  } else {
    std::string path = std::string(model_dir) + "/lyra_v1.lyrapack";
    if (!pk.open(path, &err)) {
      std::vector<uint8_t> conv;
      std::string err2;
      if (!pack_from_tflite_dir(model_dir, &conv, &err2) || !pk.adopt(std::move(conv), &err2))
        return fail(nullptr, LYRA_HIP_EMODEL, "%s; %s", err.c_str(), err2.c_str());
    }
  }
  lyra_hip_ctx* c = new lyra_hip_ctx();
  c->device = device;
  c->max_streams = max_streams;
  c->mode = requant_mode;
  if (!build_model(pk, requant_mode, &c->model, &err)) {
    delete c;
    return fail(nullptr, LYRA_HIP_EMODEL, "%s", err.c_str());
  }
  auto bail = [&](int code, const char* what) {
    std::string msg = what;
    lyra_hip_destroy(c);
    return fail(nullptr, code, "%s", msg.c_str());
  };
  {
    const char* ev = getenv("LYRA_HIP_SUBBATCHES");
    int ns = ev ? atoi(ev) : 1;  // measured on MI355X at B = 4096: 1 -> 488 us/step, 2 -> 548, 4 -> 737
    c->nsub = ns < 1 ? 1 : (ns > lyra_hip_ctx::KMAX ? lyra_hip_ctx::KMAX : ns);
  }
  // The internal events only order work of this device's streams against each other: no system-scope fence (cache
  // write-back) when they are recorded -- a recorded event costs ~5.5 us of stream bubble with it (rocprofv3 timeline).
  // Stream priorities.  Rounds 2-3 ran the decoder chain at the highest priority: the quantizer then was a 9 M-instruction
  // vector kernel that must not run beside decoder stage 0.  With the screened quantizer (round 4: 1.5 M instructions, one
  // wavefront per 16 frames) that reason is gone, and the decoder-first schedule turned out BIMODAL -- 0.290 or 0.305 ms
  // per step at B = 4096 from run to run on one box, depending on which chain ends up waiting for the other -- while the
  // small quantizer at the highest priority and both chains equal gives 0.290-0.292 every time (profiles/history/r04_prio_ab2.txt,
  // r04_prio_ab4.txt; the decoder chain one level up is bimodal again: r04_prio_ab5.txt).  The price: a blocking decode call
  // no longer overtakes an encode running beside it -- BatchLyraEncoder + BatchLyraDecoder on two host threads 6.4 M
  // frames/s instead of 7.2 M (LYRA_HIP_PRIO=0,2,0 restores the old schedule for such a service).
  int prio_lo = 0, prio_hi = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
  int prio[3] = {prio_lo, prio_lo, getenv("LYRA_HIP_FLAT_PRIO") ? prio_lo : prio_hi};
  if (const char* p = getenv("LYRA_HIP_PRIO")) {   // experiment hook: "e,d,q" each 0 = lowest .. 2 = highest
    int v[3] = {0, 0, 2};
    sscanf(p, "%d,%d,%d", &v[0], &v[1], &v[2]);
    for (int i = 0; i < 3; ++i) prio[i] = v[i] >= 2 ? prio_hi : (v[i] == 1 ? (prio_lo + prio_hi) / 2 : prio_lo);
  }
  const unsigned evflags = hipEventDisableTiming | (getenv("LYRA_HIP_EVENT_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);
  // CU partitioning (placement_probe.hip (a probe of an earlier round, removed since: git history), profiles/history/r04_placement_probe.txt): two concurrent dispatches of <= 256
  // workgroups are placed independently of each other -- of 2 x 128 workgroups 68 CUs get two and 68 none -- and a stage
  // kernel lasts as long as its slowest tile.  Streams created with complementary CU masks keep the chains apart.
  // The pattern is 32 bits repeated over the chip's CU mask; 0x00ff00ff / 0xff00ff00 give each side half of every XCD
  // whether the runtime numbers the mask bits XCC-major or interleaved (probe: 2 x 128 workgroups on 256 distinct CUs).
  // Measured (profiles/history/r04_cumask_batch_sweep.txt): +10-14 % at 256-1,024 streams, +2-5 % at 2,048 (whole tiles per CU on
  // both halves), -4...-17 % at 1,536 and from 2,560 up (each chain then wants the whole chip in turn).  The choice is made
  // once, from max_streams: a context for at most 1,024 streams runs the extractor + quantizer on one half of every XCD
  // and the decoder + noise estimator on the other.  A masked stream has no priority (the chains no longer compete) and,
  // created through hipExtStreamCreateWithCUMask, is a blocking stream with respect to the NULL stream.
  if (max_streams <= 1024) { c->cu_pat[0] = c->cu_pat[2] = 0x00ff00ffu; c->cu_pat[1] = c->cu_pat[3] = 0xff00ff00u; }
  if (const char* m = getenv("LYRA_HIP_CU_MASKS")) {   // "e,d,q,n" hex patterns; "0": none
    c->cu_pat[0] = c->cu_pat[1] = c->cu_pat[2] = c->cu_pat[3] = 0;
    sscanf(m, "%x,%x,%x,%x", &c->cu_pat[0], &c->cu_pat[1], &c->cu_pat[2], &c->cu_pat[3]);
  }
  auto make_stream = [&](hipStream_t* s, int kind, int priority) { return make_stream_kind(c, s, kind, priority); };
  for (int k = 0; k < c->nsub; ++k)
    if (make_stream(&c->se[k], 0, prio[0]) != hipSuccess ||
        make_stream(&c->sd[k], 1, prio[1]) != hipSuccess ||
        make_stream(&c->sq[k], 2, prio[2]) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_encs[0][k], evflags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_encs[1][k], evflags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_encs[2][k], evflags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_feat[k], evflags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dec[0][k], evflags) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dec[1][k], evflags) != hipSuccess)
      return bail(LYRA_HIP_EHIP, "hipStreamCreate failed");
  if (make_stream(&c->sn, 3, prio_lo) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_noise[0], evflags) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_noise[1], evflags) != hipSuccess ||
      [&] { for (auto& e : c->ev_rs_in) if (hipEventCreateWithFlags(&e, evflags) != hipSuccess) return true; return false; }() ||
      hipEventCreateWithFlags(&c->ev_ahead_order, evflags) != hipSuccess ||
      (c->nsub > 1 && [&] { for (int k = 0; k < c->nsub; ++k) if (hipEventCreateWithFlags(&c->ev_se_last[k], evflags) != hipSuccess) return true; return false; }()) ||
      hipEventCreateWithFlags(&c->ev_ahead_last, evflags) != hipSuccess)
    return bail(LYRA_HIP_EHIP, "hipStreamCreate failed");
  // A process's streams share a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and a stream gets its queue
  // when it is first used: touched here in a fixed order, the context's four streams land on four DIFFERENT queues
  // whatever order the caller uses them in later (round 6: with the queues handed out in order of first use the pipelined
  // encode ran at 16.8 or at 11.9 M frames/s depending on which calls the process had made before).
  {
    hipStream_t order[4] = {c->se[0], c->sd[0], c->sq[0], c->sn};
    for (hipStream_t s : order)
      if (hipEventRecord(c->ev_ahead_last, s) != hipSuccess) return bail(LYRA_HIP_EHIP, "hipEventRecord failed");
    if (hipEventSynchronize(c->ev_ahead_last) != hipSuccess) return bail(LYRA_HIP_EHIP, "hipEventSynchronize failed");
  }
  if (hipMalloc((void**)&c->d_state, (size_t)max_streams * st::BYTES) != hipSuccess)
    return bail(LYRA_HIP_ENOMEM, "hipMalloc(state) failed");
  if (hipMalloc((void**)&c->d_rvq_stats, 4 * sizeof(unsigned)) != hipSuccess || hipMemset(c->d_rvq_stats, 0, 4 * sizeof(unsigned)) != hipSuccess)
    return bail(LYRA_HIP_ENOMEM, "hipMalloc(rvq stats) failed");
  {
    size_t off = 0;
    for (int r = 0; r < st::R_COUNT; ++r) {   // region sizes are multiples of 256 bytes: every base stays aligned
      c->sm.base[r] = c->d_state + off;
      c->sm.bytes[r] = st::REGION_BYTES[r];
      off += (size_t)max_streams * st::REGION_BYTES[r];
    }
  }
  {
    const char* names[6] = {"ENC_S0", "ENC_S1", "ENC_S2", "DEC_S0", "DEC_S1", "DEC_S2"};
    for (int i = 0; i < 6; ++i) c->lds_pad[i] = lds_pad(names[i]);
    for (int i = 0; i < 6; ++i) c->tile_div[i] = tile_div(names[i]);
  }
  if (set_lds(enc_s0_kernel, enc_s0_lds_bytes() + c->lds_pad[0]) != hipSuccess || set_lds(enc_s1_kernel, enc_s1_lds_bytes() + c->lds_pad[1]) != hipSuccess ||
      set_lds(enc_s2_kernel, enc_s2_lds_bytes() + c->lds_pad[2]) != hipSuccess ||
#ifdef LYRA_PARKED
      set_lds(enc_side_kernel, enc_side_lds_bytes()) != hipSuccess ||
      set_lds(enc_side_dr_kernel, enc_side_lds_bytes()) != hipSuccess || set_lds(dec_side_kernel, dec_side_lds_bytes()) != hipSuccess ||
      set_lds(dec_side_dr_kernel, dec_side_lds_bytes()) != hipSuccess ||
      set_lds(enc_side_xn_kernel, enc_side_lds_bytes()) != hipSuccess || set_lds(dec_side_xn_kernel, dec_side_lds_bytes()) != hipSuccess ||
      set_lds(enc_s12_xn_kernel, enc_s12_lds_bytes()) != hipSuccess || set_lds(dec_s01_xn_kernel, dec_s01_lds_bytes()) != hipSuccess ||
#endif
      set_lds(dec_s0_kernel, dec_s0_lds_bytes() + c->lds_pad[3]) != hipSuccess ||
      set_lds(enc_s2_dr_kernel, enc_s2_lds_bytes()) != hipSuccess || set_lds(dec_s0_dr_kernel, dec_s0_lds_bytes()) != hipSuccess ||
      set_lds(enc_s2_bm_kernel, enc_s2_lds_bytes()) != hipSuccess || set_lds(dec_s0_bm_kernel, dec_s0_lds_bytes()) != hipSuccess ||
      set_lds(enc_s2_xn_kernel, enc_s2_lds_bytes() + c->lds_pad[2]) != hipSuccess || set_lds(dec_s0_xn_kernel, dec_s0_lds_bytes() + c->lds_pad[3]) != hipSuccess ||
      set_lds(dec_s1_kernel, dec_s1_lds_bytes() + c->lds_pad[4]) != hipSuccess || set_lds(dec_s2_kernel, dec_s2_lds_bytes() + c->lds_pad[5]) != hipSuccess ||
      set_lds(logmel_kernel, logmel_lds_bytes()) != hipSuccess || set_lds(cng_kernel, cng_lds_bytes()) != hipSuccess)
    return bail(LYRA_HIP_EHIP, "hipFuncSetAttribute(dynamic LDS) failed");
  for (int i = 0; i < K_COUNT; ++i) c->cw[i] = code_warm_bytes(kKernelNames[i]);
  if (c->mode == 1) {
    c->cw[K_ENC_S2] = code_warm_bytes("enc_s2_dr_kernel"); c->cw[K_DEC_S0] = code_warm_bytes("dec_s0_dr_kernel");
    c->cw[K_ENC_SIDE] = code_warm_bytes("enc_side_dr_kernel"); c->cw[K_DEC_SIDE] = code_warm_bytes("dec_side_dr_kernel");
  } else if (c->mode == 3) {
    c->cw[K_ENC_S2] = code_warm_bytes("enc_s2_bm_kernel"); c->cw[K_DEC_S0] = code_warm_bytes("dec_s0_bm_kernel");
  } else if (c->mode == 2) {
    c->cw[K_ENC_S2] = code_warm_bytes("enc_s2_xn_kernel"); c->cw[K_DEC_S0] = code_warm_bytes("dec_s0_xn_kernel");
    c->cw[K_ENC_SIDE] = code_warm_bytes("enc_side_xn_kernel"); c->cw[K_DEC_SIDE] = code_warm_bytes("dec_side_xn_kernel");
  }
#ifdef LYRA_PARKED   // parked experiments: only in the `make EXTRA=-DLYRA_PARKED` variant
  if (const char* f = getenv("LYRA_HIP_FUSED")) c->fused = c->mode == 3 ? 0 : atoi(f);   // (the parked kernels predate mode 3)
  if (const char* f = getenv("LYRA_HIP_RVQ_WIDE")) c->rvq_wide = atoi(f);
#else
  if (getenv("LYRA_HIP_FUSED") && atoi(getenv("LYRA_HIP_FUSED")))
    return bail(LYRA_HIP_EINVAL, "LYRA_HIP_FUSED: the one-launch-per-side kernels are not in this build (make EXTRA=-DLYRA_PARKED)");
#endif
  if (!(getenv("LYRA_HIP_NO_ZEROCOPY") && atoi(getenv("LYRA_HIP_NO_ZEROCOPY"))) &&
      hipHostMalloc((void**)&c->h_zc, lyra_hip_ctx::ZC_BYTES, hipHostMallocDefault) != hipSuccess)
    c->h_zc = nullptr;   // optional: the copy-engine path remains
  for (int k = 0; k < c->nsub; ++k)
    if (enc_side_done(c, k) != 0) return bail(LYRA_HIP_EHIP, "hipEventRecord failed");
  *out = c;
  int rc = lyra_hip_reset_streams(c, nullptr, 0);
  if (rc == 0) rc = sync_all(c);
  if (rc != 0) { *out = nullptr; return bail(rc, "initial state reset failed"); }
  return 0;
}

void lyra_hip_destroy(lyra_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)sync_all(c);
  twin_free(c);
  pipe_free(c);
  if (c->h_zc) (void)hipHostFree(c->h_zc);
  c->h_zc = nullptr;
  free_scratch(c);
  for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->ev_caller) (void)hipEventDestroy(c->ev_caller);
  for (int k = 0; k < lyra_hip_ctx::KMAX; ++k) {
    for (int i = 0; i < 3; ++i)
      if (c->ev_encs[i][k]) (void)hipEventDestroy(c->ev_encs[i][k]);
    if (c->ev_feat[k]) (void)hipEventDestroy(c->ev_feat[k]);
    if (c->ev_dec[0][k]) (void)hipEventDestroy(c->ev_dec[0][k]);
    if (c->ev_dec[1][k]) (void)hipEventDestroy(c->ev_dec[1][k]);
    if (c->se[k]) (void)hipStreamDestroy(c->se[k]);
    if (c->sd[k]) (void)hipStreamDestroy(c->sd[k]);
    if (c->sq[k]) (void)hipStreamDestroy(c->sq[k]);
  }
  for (hipEvent_t e : c->ev_noise)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_rs_in)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_ahead_order) (void)hipEventDestroy(c->ev_ahead_order);
  for (hipEvent_t e : c->ev_se_last)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_ahead_last) (void)hipEventDestroy(c->ev_ahead_last);
  if (c->sn) (void)hipStreamDestroy(c->sn);
  if (c->d_state) (void)hipFree(c->d_state);
  if (c->d_rvq_stats) (void)hipFree(c->d_rvq_stats);
  free_model(&c->model);
  delete c;
}

const char* lyra_hip_last_error(const lyra_hip_ctx* c) {
  if (c) return c->err.c_str();
  std::lock_guard<std::mutex> l(g_create_mu);
  return g_create_error.c_str();
}

void* lyra_hip_stream(lyra_hip_ctx* c) { return c ? (void*)c->se[0] : nullptr; }
void* lyra_hip_stream_decode(lyra_hip_ctx* c) { return c ? (void*)c->sd[0] : nullptr; }
void* lyra_hip_stream_quantizer(lyra_hip_ctx* c) { return c ? (void*)c->sq[0] : nullptr; }
void* lyra_hip_stream_noise(lyra_hip_ctx* c) { return c ? (void*)c->sn : nullptr; }
int lyra_hip_synchronize(lyra_hip_ctx* c) {
  if (!c) return LYRA_HIP_EINVAL;
  return sync_all(c);
}
size_t lyra_hip_state_bytes_per_stream(void) { return (size_t)st::BYTES; }
int lyra_hip_max_streams(const lyra_hip_ctx* c) { return c ? c->max_streams : 0; }

int lyra_hip_reset_streams(lyra_hip_ctx* c, const int32_t* ids, int n) {
  if (!c) return LYRA_HIP_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = sync_all(c);  // the reset touches encoder and decoder state: nothing may be in flight
  if (rc) return rc;
  if (!ids) {
    hipLaunchKernelGGL(reset_kernel, dim3(c->max_streams), dim3(256), 0, c->se[0], c->model.d_reset,
                       (const int32_t*)nullptr, c->max_streams, 1, c->sm);
    HIPCHK(c, hipGetLastError());
    return sync_all(c);
  }
  if ((rc = check_batch(c, n))) return rc;
  if ((rc = check_ids_host(c, ids, n))) return rc;
  if ((rc = ensure_scratch(c, n))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, c->se[0]));
  hipLaunchKernelGGL(reset_kernel, dim3(n), dim3(256), 0, c->se[0], c->model.d_reset, (const int32_t*)c->d_ids, n, 0,
                     c->sm);
  HIPCHK(c, hipGetLastError());
  return sync_all(c);
}

// ---- device-pointer variants (asynchronous; split into nsub independent sub-batches) --------------------------
int lyra_hip_extract_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm, float* d_feat) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if (!d_ids || !d_pcm || !d_feat) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  const int nk = chunks_for(c, B);
  for (int k = 0; k < nk && !rc; ++k) {
    int lo = 0, n = B;
    if (nk > 1) chunk_of(c, B, k, &lo, &n);
    if (n <= 0) continue;
    if ((rc = enc_side_begin(c, k, nk))) break;
    rc = launch_extract(c, k, lo, d_ids + lo, n, d_pcm + (size_t)lo * 320, d_feat + (size_t)lo * 64);
    if (!rc) rc = enc_side_done(c, k, nk);
  }
  c->enc_last_nk = nk;
  return rc;
}

int lyra_hip_rvq_encode_dev(lyra_hip_ctx* c, int B, const float* d_feat, int num_bits, int32_t* d_idx) {
  if (!c) return LYRA_HIP_EINVAL;
  int rc = check_bits(c, num_bits);
  if (rc) return rc;
  if (B <= 0 || !d_feat || !d_idx) return fail(c, LYRA_HIP_EINVAL, "bad batch or null pointer");
  DEVSCOPE(c);
  if ((rc = enc_side_begin(c, 0))) return rc;
  rc = launch_rvq_encode(c, 0, B, d_feat, num_bits / 4, d_idx, nullptr);
  if (!rc) rc = enc_side_done(c, 0);
  return rc;
}

int lyra_hip_rvq_decode_dev(lyra_hip_ctx* c, int B, const int32_t* d_idx, float* d_feat) {
  if (!c) return LYRA_HIP_EINVAL;
  if (B <= 0 || !d_feat || !d_idx) return fail(c, LYRA_HIP_EINVAL, "bad batch or null pointer");
  DEVSCOPE(c);
  int rc = dec_side_begin(c, 0);
  if (rc) return rc;
  rc = launch_rvq_decode(c, 0, B, d_idx, nullptr, 46, d_feat);
  if (!rc) rc = dec_side_done(c, 0, 1);
  c->n_dec_calls++;
  return rc;
}

int lyra_hip_generate_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const float* d_feat, int16_t* d_pcm) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if (!d_ids || !d_pcm || !d_feat) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  const int nk = chunks_for(c, B);
  for (int k = 0; k < nk && !rc; ++k) {
    int lo = 0, n = B;
    if (nk > 1) chunk_of(c, B, k, &lo, &n);
    if (n <= 0) continue;
    if ((rc = dec_side_begin(c, k, nk))) break;
    rc = launch_generate(c, k, lo, d_ids + lo, n, d_feat + (size_t)lo * 64, d_pcm + (size_t)lo * 320);
    if (!rc) rc = dec_side_done(c, k, nk);
  }
  c->n_dec_calls++;
  return rc;
}

int lyra_hip_logmel_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm, float* d_mel) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if (!d_ids || !d_pcm || !d_mel) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = dec_side_begin(c, 0))) return rc;
  rc = launch_logmel(c, d_ids, B, d_pcm, d_mel);
  if (!rc) rc = dec_side_done(c, 0, 1);
  c->n_dec_calls++;
  return rc;
}

int lyra_hip_encode_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm, int num_bits,
                        uint8_t* d_packets) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!d_ids || !d_pcm || !d_packets) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  const int nbytes = (num_bits + 7) / 8;
  const int nk = chunks_for(c, B);
  for (int k = 0; k < nk && !rc; ++k) {
    int lo = 0, n = B;
    if (nk > 1) chunk_of(c, B, k, &lo, &n);
    if (n <= 0) continue;
    if ((rc = encq_begin(c, k, nk))) break;
    float* feat = encq_features(c) + (size_t)lo * 64;
    rc = launch_extract(c, k, lo, d_ids + lo, n, d_pcm + (size_t)lo * 320, feat, encq_buffer_free(c, k, nk));
    if (!rc) rc = encq_handoff(c, k);
    if (!rc) rc = launch_rvq_encode(c, k, n, feat, num_bits / 4, nullptr, d_packets + (size_t)lo * nbytes, nullptr, nullptr,
                                    true);
    if (!rc) rc = encq_done(c, k);
  }
  c->encq_nk[c->n_encq_calls & 1] = nk;
  c->n_encq_calls++;
  c->enc_last_nk = nk;
  return rc;
}

int lyra_hip_decode_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const uint8_t* d_packets, int num_bits,
                        int16_t* d_pcm) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!d_ids || !d_pcm || !d_packets) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  const int nbytes = (num_bits + 7) / 8;
  const int nk = chunks_for(c, B);
  for (int k = 0; k < nk && !rc; ++k) {
    int lo = 0, n = B;
    if (nk > 1) chunk_of(c, B, k, &lo, &n);
    if (n <= 0) continue;
    if ((rc = dec_side_begin(c, k, nk))) break;
    rc = launch_generate(c, k, lo, d_ids + lo, n, nullptr, d_pcm + (size_t)lo * 320,
                         d_packets + (size_t)lo * nbytes, num_bits / 4);
    if (!rc) rc = dec_side_done(c, k, nk);
  }
  c->n_dec_calls++;
  return rc;
}

// host-pointer entry points: validate, bind the device, size the scratch, drain both streams
#define PROLOGUE(c, B)                      \
  int rc = check_batch(c, B);               \
  if (rc) return rc;                        \
  HIPCHK(c, hipSetDevice(c->device));       \
  if ((rc = ensure_scratch(c, B))) return rc; \
  if ((rc = sync_all(c))) return rc

// ---- Resampler / ComfortNoiseGenerator ------------------------------------------------------------------
// Kaiser-windowed-sinc polyphase table; the same construction as oracle/lyra_oracle.c lo_resampler_design
// (audio_dsp::QResampler restated: cutoff 0.9 x the lower Nyquist, Kaiser beta 6, radius 17 input samples, unit DC gain).
static double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k < 64; ++k) { term *= (x / (2.0 * k)) * (x / (2.0 * k)); sum += term; if (term < 1e-18 * sum) break; }
  return sum;
}
static bool resample_design(int in_rate, int out_rate, ResampleP* P) {
  static const int kRates[] = {8000, 16000, 32000, 48000};   // kSupportedSampleRates (lyra_config.h)
  bool ok_in = false, ok_out = false;
  for (int r : kRates) { ok_in |= r == in_rate; ok_out |= r == out_rate; }
  if (!ok_in || !ok_out || (in_rate != 16000 && out_rate != 16000)) return false;
  int a = in_rate, b = out_rate;
  while (b) { int t = a % b; a = b; b = t; }
  P->up = out_rate / a; P->down = in_rate / a;
  const double PI = 3.14159265358979323846;
  const int radius = st::RS_RADIUS, taps = st::RS_TAPS;
  const double cutoff = 0.9 * 0.5 * (in_rate < out_rate ? in_rate : out_rate);
  const double wc = 2.0 * cutoff / in_rate, beta = 6.0, i0b = bessel_i0(beta);
  memset(P->coef, 0, sizeof P->coef);
  for (int p = 0; p < P->up; ++p) {
    double h[64], sum = 0.0;
    for (int j = 0; j < taps; ++j) {
      const double x = (double)(radius - j) + (double)p / P->up;
      double v = 0.0;
      if (fabs(x) <= radius) {
        const double arg = PI * wc * x;
        const double sinc = fabs(arg) < 1e-12 ? 1.0 : sin(arg) / arg;
        const double y = x / radius;
        v = wc * sinc * bessel_i0(beta * sqrt(1.0 - y * y)) / i0b;
      }
      h[j] = v; sum += v;
    }
    for (int j = 0; j < taps; ++j) P->coef[p][j] = (float)(h[j] / sum);
  }
  return true;
}

// side 0: the encoder's resampler slot (external rate -> 16 kHz, encode-side stream); 1: the decoder's (16 kHz -> external)
// Work launched AHEAD on sq[0] by run_steps (see resample_in_ahead below).
// ahead_begin: first launch of a run_steps call -- after whatever se[0] did to the same slots before; ahead_end: the next
// per-call encoder-side launch that touches those slots on se[0] waits for it (wait_ahead).
static int ahead_begin(lyra_hip_ctx* c) {
  HIPCHK(c, hipEventRecord(c->ev_ahead_order, c->se[0]));
  HIPCHK(c, hipStreamWaitEvent(c->sq[0], c->ev_ahead_order, 0));
  return 0;
}
static int ahead_end(lyra_hip_ctx* c) {
  HIPCHK(c, hipEventRecord(c->ev_ahead_last, c->sq[0]));
  c->ahead_unseen = true;
  return 0;
}
static int wait_ahead(lyra_hip_ctx* c) {
#ifdef LYRA_MUTATE_NO_AHEAD_WAIT   // mutation build: tests/test_gpu_round3.py must FAIL without these edges
  return 0;
#endif
  if (c->ahead_unseen) {
    HIPCHK(c, hipStreamWaitEvent(c->se[0], c->ev_ahead_last, 0));
    c->ahead_unseen = false;
  }
  return 0;
}
// The decoder-side resampler's slots may have been touched last on the noise stream (run_steps puts the output resampler
// there, resample_deferred below): a decode-stream launch is ordered after everything enqueued on sn.
static int wait_noise_stream(lyra_hip_ctx* c) {
  if (c->n_noise_calls > c->noise_done_dec) {
    HIPCHK(c, hipStreamWaitEvent(c->sd[0], c->ev_noise[(c->n_noise_calls - 1) & 1], 0));
    if (c->nsub == 1) c->noise_done_dec = c->n_noise_calls;
  }
  return 0;
}

static int launch_resample(lyra_hip_ctx* c, int side, const int32_t* d_ids, int B, const int16_t* d_in, int n_in, int in_rate,
                    int out_rate, int16_t* d_out, int* n_out_p, int in_stride = 0, int out_stride = 0,
                    hipStream_t on_stream = nullptr) {
  ResampleP P;
  if (!resample_design(in_rate, out_rate, &P))
    return fail(c, LYRA_HIP_EINVAL, "unsupported resampling %d -> %d Hz (one side must be 16000; 8000/16000/32000/48000)", in_rate, out_rate);
  if (n_in <= 0 || n_in > 960 || n_in % P.down != 0)
    return fail(c, LYRA_HIP_EINVAL, "resample: %d input samples per stream (must be 1..960 and a multiple of %d)", n_in, P.down);
  const int n_out = n_in * P.up / P.down;
  if (n_out > 960) return fail(c, LYRA_HIP_EINVAL, "resample: %d output samples per stream exceed 960", n_out);
  hipStream_t st_ = on_stream ? on_stream : side == 0 ? c->se[0] : c->sd[0];
#ifdef LYRA_MUTATE_NO_AHEAD_WAIT
  c->rs_sn_pending = false;
#endif
  if (side == 1 && !on_stream && c->rs_sn_pending) {   // (same slots; other noise-stream work does not touch them)
    int rc = wait_noise_stream(c);
    if (rc) return rc;
    c->rs_sn_pending = false;
  }
  { ProfScope ps(c, K_RESAMPLE, st_);
    static const size_t pad = lds_pad("resample");   // experiment hook
    hipLaunchKernelGGL(resample_kernel, dim3(cdiv(B, resample_streams_per_wg())), dim3(256), resample_lds_bytes(n_in) + pad, st_, P, d_ids, B,
                       c->sm.base[side == 0 ? st::R_RS_E : st::R_RS_D], d_in, n_in, in_stride > 0 ? in_stride : n_in, d_out,
                       n_out, out_stride > 0 ? out_stride : n_out); }
  HIPCHK(c, hipGetLastError());
  if (n_out_p) *n_out_p = n_out;
  return 0;
}

static int launch_cng(lyra_hip_ctx* c, const int32_t* d_ids, int B, const float* d_features, int16_t* d_pcm) {
  // reads the decoder-side noise estimate: after every decoder-side `_dev` noise call (they run on sn)
  if (!d_features) { int rc = wait_noise_stream(c); if (rc) return rc; }
  { ProfScope ps(c, K_CNG, c->sd[0]);
    hipLaunchKernelGGL(cng_kernel, dim3(B), dim3(256), cng_lds_bytes(), c->sd[0], c->model.d_mel, c->cng_seed, d_ids, B,
                       c->sm.base[st::R_CNG], (const uint8_t*)c->sm.base[st::R_NOISE_D], d_features, d_pcm); }
  HIPCHK(c, hipGetLastError());
  return 0;
}

int lyra_hip_resample_dev(lyra_hip_ctx* c, int side, const int32_t* d_ids, int B, const int16_t* d_in, int n_in,
                          int in_rate, int out_rate, int16_t* d_out) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((side != 0 && side != 1) || !d_ids || !d_in || !d_out) return fail(c, LYRA_HIP_EINVAL, "bad side or null pointer");
  DEVSCOPE(c);
  if (side == 0) {
    if ((rc = enc_side_begin(c, 0))) return rc;
    if ((rc = wait_ahead(c))) return rc;
    rc = launch_resample(c, 0, d_ids, B, d_in, n_in, in_rate, out_rate, d_out, nullptr);
    if (!rc) rc = enc_side_done(c, 0);
    return rc;
  }
  if ((rc = dec_side_begin(c, 0))) return rc;
  rc = launch_resample(c, 1, d_ids, B, d_in, n_in, in_rate, out_rate, d_out, nullptr);
  if (!rc) rc = dec_side_done(c, 0, 1);
  c->n_dec_calls++;
  return rc;
}

// run_steps' output resampler (lyra_decoder.cc:107-113): like the decoder-side noise estimator it only consumes the hop
// the decoder has just written, so it runs on the noise stream behind the decoder's last stage, underneath the next step,
// instead of lengthening the decoder's chain.  Same bookkeeping as a decoder-side noise call: its input obeys the
// two-buffer rule through dec_side_begin's wait for all noise-stream calls but the most recent one.
static int resample_deferred(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_in, int n_in, int in_rate,
                      int out_rate, int16_t* d_out) {
  DEVSCOPE(c);
  int rc = noise_dev_begin(c);
  if (rc) return rc;
  rc = launch_resample(c, 1, d_ids, B, d_in, n_in, in_rate, out_rate, d_out, nullptr, 0, 0, c->sn);
  c->rs_sn_pending = true;
  if (!rc) rc = noise_dev_done(c);
  return rc;
}

// run_steps with BOTH decoder-side legs (NoiseEstimator on every decoded hop + output resampler): the two launches of a hop are
// ONE noise-stream call -- one begin, one record.  The two-buffer rule counts calls ("all noise-stream calls but the most
// recent one"): counted separately, the quantizer of step i waited for the ESTIMATOR OF STEP i-1, so the decoder chain of
// step i could not start before the previous hop's estimator + this hop's quantizer had run behind the previous decoder chain
// (rocprofv3 timeline: 126 us of nothing on the decoder stream per step, 0.422 ms per step; profiles/r06_modes_timelines.txt).
static int noise_and_resample_deferred(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm16, int32_t* d_is_noise,
                                       int out_rate, int16_t* d_out) {
  DEVSCOPE(c);
  int rc = ensure_scratch(c, B);
  if (rc) return rc;
  if ((rc = noise_dev_begin(c))) return rc;
  rc = launch_noise(c, 1, c->sn, d_ids, B, d_pcm16, d_is_noise, nullptr);
  if (!rc) rc = launch_resample(c, 1, d_ids, B, d_pcm16, 320, 16000, out_rate, d_out, nullptr, 0, 0, c->sn);
  c->rs_sn_pending = true;
  if (!rc) rc = noise_dev_done(c);
  return rc;
}

// Work run_steps launches AHEAD on the quantizer stream sq[0]: the encoder's input resampler (lyra_encoder.cc:119-122)
// only depends on the caller's input ring and on slots nothing else touches, so the hop of step i+2 is resampled in front
// of rvq_encode(i), underneath step i's feature extractor instead of lengthening the extractor's chain, into buffer
// (i+2) mod 3; the extractor of step i+2 waits for its event.  The buffer it overwrites was last read by the extractor
// of step i-1, which rvq_encode(i-1) -- earlier on the same stream -- has waited for.  TWO hops ahead since round 6: at an
// external rate the decoder chain is the slower one, the quantizer runs late (it waits for the noise stream's work of two
// hops ago) and a hop resampled only ONE step ahead, queued behind it, arrived after the extractor wanted it -- a ~90 us
// stall every second hop (profiles/r06_modes_timelines.txt; LYRA_HIP_RS_LEAD=1 is the old form).  (The DTX NoiseEstimator
// was tried on this stream too: no gain -- +31 us per step in front of the extractor or ahead -- and not kept.)
static int resample_in_ahead(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_in, int n_in, int in_rate, long step) {
  const int p = (int)(step % lyra_hip_ctx::RS_RING);
  int rc = launch_resample(c, 0, d_ids, B, d_in, n_in, in_rate, 16000, c->d_rs16[p], nullptr, 0, 0, c->sq[0]);
  if (rc) return rc;
  HIPCHK(c, hipEventRecord(c->ev_rs_in[p], c->sq[0]));
  return ahead_end(c);
}
int lyra_hip_resample(lyra_hip_ctx* c, int side, const int32_t* ids, int B, const int16_t* in, int n_in, int in_rate,
                      int out_rate, int16_t* out) {
  PROLOGUE(c, B);
  if ((side != 0 && side != 1) || !in || !out) return fail(c, LYRA_HIP_EINVAL, "bad side or null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  if (n_in <= 0 || n_in > 960) return fail(c, LYRA_HIP_EINVAL, "resample: %d input samples per stream (1..960)", n_in);
  hipStream_t st_ = side == 0 ? c->se[0] : c->sd[0];
  int32_t* dids = side == 0 ? c->d_ids : c->d_ids_dec;
  HIPCHK(c, hipMemcpyAsync(dids, ids, (size_t)B * 4, hipMemcpyHostToDevice, st_));
  HIPCHK(c, hipMemcpyAsync(c->d_rs_in, in, (size_t)B * n_in * 2, hipMemcpyHostToDevice, st_));
  int n_out = 0;
  if (side == 0 && (rc = wait_ahead(c))) return rc;
  if ((rc = launch_resample(c, side, dids, B, c->d_rs_in, n_in, in_rate, out_rate, c->d_rs_out, &n_out))) return rc;
  HIPCHK(c, hipMemcpyAsync(out, c->d_rs_out, (size_t)B * n_out * 2, hipMemcpyDeviceToHost, st_));
  if (side == 0 && (rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(st_));
  return 0;
}

int lyra_hip_comfort_noise_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const float* d_features, int16_t* d_pcm) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if (!d_ids || !d_pcm) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = dec_side_begin(c, 0))) return rc;
  rc = launch_cng(c, d_ids, B, d_features, d_pcm);
  if (!rc) rc = dec_side_done(c, 0, 1);
  c->n_dec_calls++;
  return rc;
}

int lyra_hip_comfort_noise(lyra_hip_ctx* c, const int32_t* ids, int B, const float* features, int16_t* pcm) {
  PROLOGUE(c, B);
  if (!pcm) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  hipStream_t st_ = c->sd[0];
  HIPCHK(c, hipMemcpyAsync(c->d_ids_dec, ids, (size_t)B * 4, hipMemcpyHostToDevice, st_));
  if (features) HIPCHK(c, hipMemcpyAsync(c->d_mel, features, (size_t)B * 160 * 4, hipMemcpyHostToDevice, st_));
  if ((rc = launch_cng(c, c->d_ids_dec, B, features ? c->d_mel : nullptr, c->d_pcm_out))) return rc;
  HIPCHK(c, hipMemcpyAsync(pcm, c->d_pcm_out, (size_t)B * 640, hipMemcpyDeviceToHost, st_));
  HIPCHK(c, hipStreamSynchronize(st_));
  return 0;
}

int lyra_hip_set_encoder_sample_rate(lyra_hip_ctx* c, int sample_rate_hz) {
  if (!c) return LYRA_HIP_EINVAL;
  if (sample_rate_hz != 8000 && sample_rate_hz != 16000 && sample_rate_hz != 32000 && sample_rate_hz != 48000)
    return fail(c, LYRA_HIP_EINVAL, "sample rate %d Hz is not supported by the codec (lyra_config.h:57)", sample_rate_hz);
  c->enc_noise_rate = sample_rate_hz;
  return 0;
}

int lyra_hip_set_cng_seed(lyra_hip_ctx* c, uint64_t seed) {
  if (!c) return LYRA_HIP_EINVAL;
  c->cng_seed = seed;
  return 0;
}

// ---- NoiseEstimator / DTX ---------------------------------------------------------------------------
int lyra_hip_noise_receive_dev(lyra_hip_ctx* c, int side, const int32_t* d_ids, int B, const int16_t* d_pcm,
                               int32_t* d_is_noise) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((side != 0 && side != 1) || !d_ids || !d_pcm || !d_is_noise) return fail(c, LYRA_HIP_EINVAL, "bad side or null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  if (side == 0) {
    if ((rc = enc_side_begin(c, 0))) return rc;
    rc = launch_noise(c, 0, c->se[0], d_ids, B, d_pcm, d_is_noise, nullptr);
    if (!rc) rc = enc_side_done(c, 0);
    return rc;
  }
  if ((rc = noise_dev_begin(c))) return rc;
  rc = launch_noise(c, 1, c->sn, d_ids, B, d_pcm, d_is_noise, nullptr);
  if (!rc) rc = noise_dev_done(c);
  return rc;
}

int lyra_hip_encode_dtx_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm, int num_bits,
                            uint8_t* d_packets, int32_t* d_packet_bytes) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!d_ids || !d_pcm || !d_packets || !d_packet_bytes) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  DEVSCOPE(c);
  if ((rc = ensure_scratch(c, B))) return rc;
  if ((rc = encq_begin(c, 0))) return rc;
  // lyra_encoder.cc:131-141: the noise estimator sees every hop; only non-noise hops reach the feature extractor
  // (d_live_ids / d_flag_enc are rewritten by the next call's noise kernel on se[0]: it must not overtake this call's
  // quantizer, which reads d_live_ids on sd[0] -- the next call's encq_begin waits for the call before the previous
  // one only, so the mask travels with the features: one buffer per parity)
  float* feat = encq_features(c);
  int32_t* live = (c->n_encq_calls & 1) ? c->d_live_ids2 : c->d_live_ids;
  {   // `live` travels with the features: both were last read by the quantizer(s) of the call before the previous one
    const EventList busy = encq_buffer_free(c, 0, 1);
    for (int i = 0; i < busy.n; ++i) HIPCHK(c, hipStreamWaitEvent(c->se[0], busy.e[i], 0));
  }
  rc = launch_noise(c, 0, c->se[0], d_ids, B, d_pcm, c->d_flag_enc, live);
  if (!rc) rc = launch_extract(c, 0, 0, live, B, d_pcm, feat);
  if (!rc) rc = encq_handoff(c, 0);
  if (!rc) rc = launch_rvq_encode(c, 0, B, feat, num_bits / 4, nullptr, d_packets, live, d_packet_bytes, true);
  if (!rc) rc = encq_done(c, 0);
  c->encq_nk[c->n_encq_calls & 1] = 1;
  c->n_encq_calls++;
  c->enc_last_nk = 1;
  return rc;
}
int lyra_hip_noise_receive(lyra_hip_ctx* c, int side, const int32_t* ids, int B, const int16_t* pcm, int32_t* is_noise) {
  PROLOGUE(c, B);
  if ((side != 0 && side != 1) || !pcm || !is_noise) return fail(c, LYRA_HIP_EINVAL, "bad side or null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  hipStream_t st_ = side == 0 ? c->se[0] : c->sd[0];
  int32_t* dids = side == 0 ? c->d_ids : c->d_ids_dec;
  int16_t* dpcm = side == 0 ? c->d_pcm_in : c->d_pcm_out;
  int32_t* dflag = side == 0 ? c->d_flag_enc : c->d_flag_dec;
  HIPCHK(c, hipMemcpyAsync(dids, ids, (size_t)B * 4, hipMemcpyHostToDevice, st_));
  HIPCHK(c, hipMemcpyAsync(dpcm, pcm, (size_t)B * 640, hipMemcpyHostToDevice, st_));
  if ((rc = launch_noise(c, side, st_, dids, B, dpcm, dflag, nullptr))) return rc;
  HIPCHK(c, hipMemcpyAsync(is_noise, dflag, (size_t)B * 4, hipMemcpyDeviceToHost, st_));
  if (side == 0 && (rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(st_));
  return 0;
}

int lyra_hip_noise_estimate(lyra_hip_ctx* c, int side, const int32_t* ids, int B, float* estimate) {
  PROLOGUE(c, B);
  if ((side != 0 && side != 1) || !estimate) return fail(c, LYRA_HIP_EINVAL, "bad side or null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  hipStream_t st_ = c->sd[0];
  HIPCHK(c, hipMemcpyAsync(c->d_ids_dec, ids, (size_t)B * 4, hipMemcpyHostToDevice, st_));
  hipLaunchKernelGGL(noise_read_kernel, dim3(cdiv(B * 160, 256)), dim3(256), 0, st_, (const int32_t*)c->d_ids_dec, B,
                     (const uint8_t*)c->sm.base[side == 0 ? st::R_NOISE_E : st::R_NOISE_D], (int)st::N_EST, c->d_mel);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(estimate, c->d_mel, (size_t)B * 160 * 4, hipMemcpyDeviceToHost, st_));
  HIPCHK(c, hipStreamSynchronize(st_));
  return 0;
}

int lyra_hip_encode_dtx(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, int num_bits, uint8_t* packets,
                        int32_t* packet_bytes) {
  PROLOGUE(c, B);
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!pcm || !packets || !packet_bytes) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  const int nbytes = (num_bits + 7) / 8;
  HIPCHK(c, hipMemcpyAsync(c->d_ids, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->se[0]));
  HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, (size_t)B * 640, hipMemcpyHostToDevice, c->se[0]));
  HIPCHK(c, hipMemsetAsync(c->d_pkt, 0, (size_t)B * nbytes, c->se[0]));   // empty packets read back as zeros
  if ((rc = launch_noise(c, 0, c->se[0], c->d_ids, B, c->d_pcm_in, c->d_flag_enc, c->d_live_ids))) return rc;
  if ((rc = launch_extract(c, 0, 0, c->d_live_ids, B, c->d_pcm_in, c->d_feat))) return rc;
  if ((rc = launch_rvq_encode(c, 0, B, c->d_feat, num_bits / 4, nullptr, c->d_pkt, c->d_live_ids, c->d_pkt_bytes))) return rc;
  HIPCHK(c, hipMemcpyAsync(packets, c->d_pkt, (size_t)B * nbytes, hipMemcpyDeviceToHost, c->se[0]));
  HIPCHK(c, hipMemcpyAsync(packet_bytes, c->d_pkt_bytes, (size_t)B * 4, hipMemcpyDeviceToHost, c->se[0]));
  if ((rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->se[0]));
  return 0;
}

// ---- ordering against a caller-owned stream (all library streams are non-blocking: they do NOT order
//      against the null stream or any other stream by themselves) -----------------------------------------------
int lyra_hip_wait_for_stream(lyra_hip_ctx* c, void* caller_stream) {
  if (!c) return LYRA_HIP_EINVAL;
  DEVSCOPE(c);
  if (!c->ev_caller) HIPCHK(c, hipEventCreateWithFlags(&c->ev_caller, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_caller, (hipStream_t)caller_stream));
  for (int k = 0; k < c->nsub; ++k) {
    HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_caller, 0));
    HIPCHK(c, hipStreamWaitEvent(c->sd[k], c->ev_caller, 0));
    HIPCHK(c, hipStreamWaitEvent(c->sq[k], c->ev_caller, 0));
  }
  HIPCHK(c, hipStreamWaitEvent(c->sn, c->ev_caller, 0));
  return 0;
}

int lyra_hip_stream_wait(lyra_hip_ctx* c, void* caller_stream) {
  if (!c) return LYRA_HIP_EINVAL;
  DEVSCOPE(c);
  if (!c->ev_caller) HIPCHK(c, hipEventCreateWithFlags(&c->ev_caller, hipEventDisableTiming));
  for (int k = 0; k < c->nsub; ++k) {
    // an event may be re-recorded once the wait that used it has been enqueued
    HIPCHK(c, hipEventRecord(c->ev_caller, c->se[k]));
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)caller_stream, c->ev_caller, 0));
    HIPCHK(c, hipEventRecord(c->ev_caller, c->sd[k]));
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)caller_stream, c->ev_caller, 0));
    HIPCHK(c, hipEventRecord(c->ev_caller, c->sq[k]));
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)caller_stream, c->ev_caller, 0));
  }
  HIPCHK(c, hipEventRecord(c->ev_caller, c->sn));
  HIPCHK(c, hipStreamWaitEvent((hipStream_t)caller_stream, c->ev_caller, 0));
  return 0;
}

// The encode-side, decode-side and quantizer streams at new priorities (0 lowest .. 2 highest; HIP fixes a stream's
// priority at creation, so the streams are drained, destroyed and created again; CU masks, events and all bookkeeping stay).
int lyra_hip_set_stream_priorities(lyra_hip_ctx* c, int enc, int dec, int quant) {
  if (!c) return LYRA_HIP_EINVAL;
  if (enc < 0 || enc > 2 || dec < 0 || dec > 2 || quant < 0 || quant > 2)
    return fail(c, LYRA_HIP_EINVAL, "stream priorities are 0 (lowest) .. 2 (highest)");
  DEVSCOPE(c);
  int rc = sync_all(c);
  if (rc) return rc;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
  auto level = [&](int v) { return v >= 2 ? hi : (v == 1 ? (lo + hi) / 2 : lo); };
  for (int k = 0; k < c->nsub; ++k) {
    hipStream_t* st3[3] = {&c->se[k], &c->sd[k], &c->sq[k]};
    const int want[3] = {level(enc), level(dec), level(quant)};
    for (int i = 0; i < 3; ++i) {
      hipStream_t fresh = nullptr;
      HIPCHK(c, make_stream_kind(c, &fresh, i, want[i]));
      (void)hipStreamDestroy(*st3[i]);
      *st3[i] = fresh;
    }
  }
  return 0;
}

int lyra_hip_set_serial(lyra_hip_ctx* c, int on) {
  if (!c) return LYRA_HIP_EINVAL;
  c->serial = on != 0;
  return 0;
}

// ---- host-pointer variants (synchronous) --------------------------------------------------------------------
int lyra_hip_extract(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, float* features) {
  PROLOGUE(c, B);
  if (!pcm || !features) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  if (c->zc(B)) {   // zero-copy: see lyra_hip_ctx::h_zc
    int32_t* z_ids = (int32_t*)(c->h_zc + lyra_hip_ctx::ZC_IDS);
    int16_t* z_in = (int16_t*)(c->h_zc + lyra_hip_ctx::ZC_IN);
    float* z_out = (float*)(c->h_zc + lyra_hip_ctx::ZC_OUT);
    std::memcpy(z_ids, ids, (size_t)B * 4);
    std::memcpy(z_in, pcm, (size_t)B * 640);
    if ((rc = launch_extract(c, 0, 0, z_ids, B, z_in, z_out))) return rc;
    if ((rc = enc_side_done(c, 0))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->se[0]));
    std::memcpy(features, z_out, (size_t)B * 256);
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_ids, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->se[0]));
  HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, (size_t)B * 640, hipMemcpyHostToDevice, c->se[0]));
  if ((rc = launch_extract(c, 0, 0, c->d_ids, B, c->d_pcm_in, c->d_feat))) return rc;
  HIPCHK(c, hipMemcpyAsync(features, c->d_feat, (size_t)B * 256, hipMemcpyDeviceToHost, c->se[0]));
  if ((rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->se[0]));
  return 0;
}

int lyra_hip_rvq_encode(lyra_hip_ctx* c, int B, const float* features, int num_bits, int32_t* indices) {
  if (!c) return LYRA_HIP_EINVAL;
  int rc = check_bits(c, num_bits);
  if (rc) return rc;
  if (B <= 0 || !features || !indices) return fail(c, LYRA_HIP_EINVAL, "bad batch or null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  if ((rc = ensure_scratch(c, B))) return rc;
  if (c->zc(B)) {
    float* z_in = (float*)(c->h_zc + lyra_hip_ctx::ZC_IN);
    int32_t* z_out = (int32_t*)(c->h_zc + lyra_hip_ctx::ZC_OUT);
    std::memcpy(z_in, features, (size_t)B * 256);
    if ((rc = launch_rvq_encode(c, 0, B, z_in, num_bits / 4, z_out, nullptr))) return rc;
    if ((rc = enc_side_done(c, 0))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->se[0]));
    std::memcpy(indices, z_out, (size_t)B * 46 * 4);
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_feat, features, (size_t)B * 256, hipMemcpyHostToDevice, c->se[0]));
  if ((rc = launch_rvq_encode(c, 0, B, c->d_feat, num_bits / 4, c->d_idx, nullptr))) return rc;
  HIPCHK(c, hipMemcpyAsync(indices, c->d_idx, (size_t)B * 46 * 4, hipMemcpyDeviceToHost, c->se[0]));
  if ((rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->se[0]));
  return 0;
}

int lyra_hip_rvq_decode(lyra_hip_ctx* c, int B, const int32_t* indices, float* features) {
  if (!c) return LYRA_HIP_EINVAL;
  if (B <= 0 || !features || !indices) return fail(c, LYRA_HIP_EINVAL, "bad batch or null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_scratch(c, B))) return rc;
  if ((rc = dec_side_begin(c, 0))) return rc;
  if (c->zc(B)) {
    int32_t* z_in = (int32_t*)(c->h_zc + lyra_hip_ctx::ZC_IN);
    float* z_out = (float*)(c->h_zc + lyra_hip_ctx::ZC_OUT);
    std::memcpy(z_in, indices, (size_t)B * 46 * 4);
    if ((rc = launch_rvq_decode(c, 0, B, z_in, nullptr, 46, z_out))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->sd[0]));
    std::memcpy(features, z_out, (size_t)B * 256);
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_idx, indices, (size_t)B * 46 * 4, hipMemcpyHostToDevice, c->sd[0]));
  if ((rc = launch_rvq_decode(c, 0, B, c->d_idx, nullptr, 46, c->d_lossy))) return rc;
  HIPCHK(c, hipMemcpyAsync(features, c->d_lossy, (size_t)B * 256, hipMemcpyDeviceToHost, c->sd[0]));
  HIPCHK(c, hipStreamSynchronize(c->sd[0]));
  return 0;
}

int lyra_hip_generate(lyra_hip_ctx* c, const int32_t* ids, int B, const float* features, int16_t* pcm) {
  PROLOGUE(c, B);
  if (!pcm || !features) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  if ((rc = dec_side_begin(c, 0))) return rc;
  if (c->zc(B)) {
    int32_t* z_ids = (int32_t*)(c->h_zc + lyra_hip_ctx::ZC_IDS);
    float* z_in = (float*)(c->h_zc + lyra_hip_ctx::ZC_IN);
    int16_t* z_out = (int16_t*)(c->h_zc + lyra_hip_ctx::ZC_OUT);
    std::memcpy(z_ids, ids, (size_t)B * 4);
    std::memcpy(z_in, features, (size_t)B * 256);
    if ((rc = launch_generate(c, 0, 0, z_ids, B, z_in, z_out))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->sd[0]));
    std::memcpy(pcm, z_out, (size_t)B * 640);
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_ids_dec, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->sd[0]));
  HIPCHK(c, hipMemcpyAsync(c->d_lossy, features, (size_t)B * 256, hipMemcpyHostToDevice, c->sd[0]));
  if ((rc = launch_generate(c, 0, 0, c->d_ids_dec, B, c->d_lossy, c->d_pcm_out))) return rc;
  HIPCHK(c, hipMemcpyAsync(pcm, c->d_pcm_out, (size_t)B * 640, hipMemcpyDeviceToHost, c->sd[0]));
  HIPCHK(c, hipStreamSynchronize(c->sd[0]));
  return 0;
}

int lyra_hip_logmel(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, float* mel) {
  PROLOGUE(c, B);
  if (!pcm || !mel) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  if ((rc = dec_side_begin(c, 0))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_ids_dec, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->sd[0]));
  HIPCHK(c, hipMemcpyAsync(c->d_pcm_out, pcm, (size_t)B * 640, hipMemcpyHostToDevice, c->sd[0]));
  if ((rc = launch_logmel(c, c->d_ids_dec, B, c->d_pcm_out, c->d_mel))) return rc;
  HIPCHK(c, hipMemcpyAsync(mel, c->d_mel, (size_t)B * 160 * 4, hipMemcpyDeviceToHost, c->sd[0]));
  HIPCHK(c, hipStreamSynchronize(c->sd[0]));
  return 0;
}

int lyra_hip_encode(lyra_hip_ctx* c, const int32_t* ids, int B, const int16_t* pcm, int num_bits, uint8_t* packets) {
  PROLOGUE(c, B);
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!pcm || !packets) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  const int nbytes = (num_bits + 7) / 8;
  HIPCHK(c, hipMemcpyAsync(c->d_ids, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->se[0]));
  HIPCHK(c, hipMemcpyAsync(c->d_pcm_in, pcm, (size_t)B * 640, hipMemcpyHostToDevice, c->se[0]));
  if ((rc = launch_extract(c, 0, 0, c->d_ids, B, c->d_pcm_in, c->d_feat))) return rc;
  if ((rc = launch_rvq_encode(c, 0, B, c->d_feat, num_bits / 4, nullptr, c->d_pkt))) return rc;
  HIPCHK(c, hipMemcpyAsync(packets, c->d_pkt, (size_t)B * nbytes, hipMemcpyDeviceToHost, c->se[0]));
  if ((rc = enc_side_done(c, 0))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->se[0]));
  return 0;
}

int lyra_hip_decode(lyra_hip_ctx* c, const int32_t* ids, int B, const uint8_t* packets, int num_bits, int16_t* pcm) {
  PROLOGUE(c, B);
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!pcm || !packets) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = check_ids_host(c, ids, B))) return rc;
  const int nbytes = (num_bits + 7) / 8;
  if ((rc = dec_side_begin(c, 0))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_ids_dec, ids, (size_t)B * 4, hipMemcpyHostToDevice, c->sd[0]));
  // decode-side packet staging reuses the idx scratch (the encode side owns d_pkt)
  uint8_t* d_pk = reinterpret_cast<uint8_t*>(c->d_idx);
  HIPCHK(c, hipMemcpyAsync(d_pk, packets, (size_t)B * nbytes, hipMemcpyHostToDevice, c->sd[0]));
  if ((rc = launch_rvq_decode(c, 0, B, nullptr, d_pk, num_bits / 4, c->d_lossy))) return rc;
  if ((rc = launch_generate(c, 0, 0, c->d_ids_dec, B, c->d_lossy, c->d_pcm_out))) return rc;
  HIPCHK(c, hipMemcpyAsync(pcm, c->d_pcm_out, (size_t)B * 640, hipMemcpyDeviceToHost, c->sd[0]));
  HIPCHK(c, hipStreamSynchronize(c->sd[0]));
  return 0;
}

// ---- one hop at an external sample rate, one call per side -----------------------------------------------------------
// LyraEncoder::Encode (lyra_encoder.cc:113-156: resampler :119-122, DTX decision :131-141, extractor, quantizer) and
// LyraDecoder::DecodeSamples for a received hop (lyra_decoder.cc: generative model, NoiseEstimator :304-311, resampler
// :107-113) each as ONE call of its side.  The results are those of the individual `_dev` calls; what differs is what the
// ordering rules of lyra_hip.h "Streams" count: issued one by one, a hop at an external rate makes two encode-side and two or
// three decode-side calls, and rule (2) -- "all decode-side calls but the most recent ONE" -- then orders the next hop's
// extractor and quantizer behind this hop's decoder chain (bench.py --per-call --rate 48000: 10.4 M frames/s against 13.2 M
// through run_steps; profiles/r06_per_call.txt).  Here the encoder's resampler runs in front of the extractor on the
// extractor's own stream (its output never leaves the library: no cross-side edge), and the decoder's estimator and
// resampler are one noise-stream call behind the decoder's last stage, as inside lyra_hip_run_steps_dev.
int lyra_hip_encode_ext_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const int16_t* d_pcm_ext, int sample_rate_hz,
                            int num_bits, int dtx, uint8_t* d_packets, int32_t* d_packet_bytes) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  if ((rc = check_bits(c, num_bits))) return rc;
  if (!d_ids || !d_pcm_ext || !d_packets || (dtx && !d_packet_bytes)) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  const int ext = sample_rate_hz;
  if (ext != 8000 && ext != 16000 && ext != 32000 && ext != 48000)
    return fail(c, LYRA_HIP_EINVAL, "sample rate %d Hz is not supported by the codec (lyra_config.h:57)", ext);
  if (dtx && c->enc_noise_rate != ext)   // as lyra_hip_run_steps_dev: the DTX estimator is created at the encoder's external rate
    return fail(c, LYRA_HIP_EINVAL, "encode_ext: DTX at %d Hz but the encoder-side noise estimator is set up for %d Hz "
                "(call lyra_hip_set_encoder_sample_rate(%d) first)", ext, c->enc_noise_rate, ext);
  const int16_t* in = d_pcm_ext;
  if (ext != 16000) {
    DEVSCOPE(c);
    if ((rc = ensure_scratch(c, B))) return rc;
    // same resampler slots as lyra_hip_resample_dev(ENCODER) and run_steps' ahead launches: behind those; the 16 kHz hop
    // goes to a library buffer that only this stream's extractor reads (the previous hop's read precedes in stream order)
    if ((rc = wait_ahead(c))) return rc;
    if ((rc = encq_begin(c, 0, 1))) return rc;   // (split contexts: after every chunk of the encode-side call before; serial mode: in call order)
    if ((rc = launch_resample(c, 0, d_ids, B, d_pcm_ext, 320 * (ext / 1000) / 16, ext, 16000, c->d_rs16[0], nullptr))) return rc;
    if (c->nsub > 1) {   // the chunks of a split encode run on se[1..]: they read what se[0] has just written
      HIPCHK(c, hipEventRecord(c->ev_ahead_order, c->se[0]));
      for (int k = 1; k < c->nsub; ++k) HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_ahead_order, 0));
    }
    in = c->d_rs16[0];
  }
  return dtx ? lyra_hip_encode_dtx_dev(c, d_ids, B, in, num_bits, d_packets, d_packet_bytes)
             : lyra_hip_encode_dev(c, d_ids, B, in, num_bits, d_packets);
}

int lyra_hip_decode_ext_dev(lyra_hip_ctx* c, const int32_t* d_ids, int B, const uint8_t* d_packets, int num_bits,
                            int sample_rate_hz, int estimate_noise, int16_t* d_pcm16, int16_t* d_pcm_ext,
                            int32_t* d_is_noise) {
  int rc = check_batch(c, B);
  if (rc) return rc;
  const int ext = sample_rate_hz;
  if (ext != 8000 && ext != 16000 && ext != 32000 && ext != 48000)
    return fail(c, LYRA_HIP_EINVAL, "sample rate %d Hz is not supported by the codec (lyra_config.h:57)", ext);
  if (!d_pcm16 || (ext != 16000 && !d_pcm_ext) || (estimate_noise && !d_is_noise)) return fail(c, LYRA_HIP_EINVAL, "null pointer");
  if ((rc = lyra_hip_decode_dev(c, d_ids, B, d_packets, num_bits, d_pcm16))) return rc;
  const bool rs = ext != 16000;
  const bool off_chain = !c->serial;   // as in lyra_hip_run_steps_dev (split contexts included)
  if (estimate_noise && rs && off_chain) return noise_and_resample_deferred(c, d_ids, B, d_pcm16, d_is_noise, ext, d_pcm_ext);
  if (estimate_noise && (rc = lyra_hip_noise_receive_dev(c, LYRA_HIP_SIDE_DECODER, d_ids, B, d_pcm16, d_is_noise))) return rc;
  if (rs)
    rc = off_chain ? resample_deferred(c, d_ids, B, d_pcm16, 320, 16000, ext, d_pcm_ext)
                   : lyra_hip_resample_dev(c, LYRA_HIP_SIDE_DECODER, d_ids, B, d_pcm16, 320, 16000, ext, d_pcm_ext);
  return rc;
}

// ---- many steps from one call ---------------------------------------------------------------------------------------
// What lyra_benchmark's loop does per hop (lyra_benchmark_lib.cc:121-160) and what LyraEncoder::Encode /
// LyraDecoder::DecodeSamples do around it (lyra_encoder.cc:113-156, lyra_decoder.cc:284-326), for n_steps hops of B
// streams, enqueued by one call: no host language in the loop.
int lyra_hip_run_steps_dev(lyra_hip_ctx* c, const lyra_hip_steps* S) {
  if (!c || !S) return LYRA_HIP_EINVAL;
  int rc = check_batch(c, S->B);
  if (rc) return rc;
  const unsigned F = S->flags;
  const bool enc = F & LYRA_HIP_STEP_ENCODE, dec = F & LYRA_HIP_STEP_DECODE, feats = S->d_features != nullptr;
  if (!enc && !dec) return fail(c, LYRA_HIP_EINVAL, "run_steps: neither ENCODE nor DECODE requested");
  if ((enc || (dec && !feats)) && (rc = check_bits(c, S->num_bits))) return rc;
  if (!S->d_stream_ids || S->n_steps < 0 || S->first_step < 0) return fail(c, LYRA_HIP_EINVAL, "run_steps: bad argument");
  if (enc && (!S->d_pcm_ring || S->ring <= 0)) return fail(c, LYRA_HIP_EINVAL, "run_steps: ENCODE needs d_pcm_ring / ring");
  if ((enc || (dec && !feats && !S->d_packet_ring)) && (!S->d_packets[0] || !S->d_packets[1]))
    return fail(c, LYRA_HIP_EINVAL, "run_steps: two packet buffers needed");
  if (dec && (!S->d_pcm_out[0] || !S->d_pcm_out[1])) return fail(c, LYRA_HIP_EINVAL, "run_steps: two PCM output buffers needed");
  if ((F & LYRA_HIP_STEP_DTX) && (!S->d_packet_bytes[0] || !S->d_packet_bytes[1]))
    return fail(c, LYRA_HIP_EINVAL, "run_steps: DTX needs two packet_bytes buffers");
  if ((F & LYRA_HIP_STEP_DECODER_NOISE) && (!dec || !S->d_is_noise))
    return fail(c, LYRA_HIP_EINVAL, "run_steps: DECODER_NOISE needs DECODE and d_is_noise");
  const int ext = S->external_rate ? S->external_rate : 16000;
  const bool rs = ext != 16000;
  // A DTX LyraEncoder created at `ext` hands that rate to its NoiseEstimator (lyra_encoder.cc:82-85): a step that claims to
  // run what LyraEncoder::Encode runs must not combine an external rate with another estimator
  if ((F & LYRA_HIP_STEP_DTX) && enc && c->enc_noise_rate != ext)
    return fail(c, LYRA_HIP_EINVAL, "run_steps: DTX at external_rate %d but the encoder-side noise estimator is set up for %d Hz "
                "(call lyra_hip_set_encoder_sample_rate(%d) first)", ext, c->enc_noise_rate, ext);
  const int n_ext = 320 * (ext / 1000) / 16;   // samples per 20 ms hop at the external rate
  if (rs) {
    if (ext != 8000 && ext != 32000 && ext != 48000) return fail(c, LYRA_HIP_EINVAL, "run_steps: external_rate %d", ext);
    if (dec && (!S->d_ext_out[0] || !S->d_ext_out[1])) return fail(c, LYRA_HIP_EINVAL, "run_steps: two external-rate output buffers needed");
    if ((rc = ensure_scratch(c, S->B))) return rc;
  }
  const size_t B = (size_t)S->B;
  // the resamplers leave the codec's chains (resample_in_ahead / resample_deferred) in the default, unsplit configuration;
  // with sub-batches or strict call order they stay where the individual calls put them
  const bool rs_off_chain = rs && !c->serial && c->nsub == 1;
  // The decoder-side legs leave the chain on split contexts too (round 6): the noise stream joins every chunk's decode
  // (noise_dev_begin) and dec_side_begin orders every chunk behind the noise stream's older work -- decode-only at 8192 streams
  // in two sub-batches, 48 kHz: 26.1 -> 27.1 M frames/s (profiles/r06_cfg4_legs.txt).  The INPUT resampler stays on the chain
  // there: ahead on sq[0] it would only be ordered behind chunk 0's extractor.  LYRA_HIP_RS_OUT_ON_CHAIN_SPLIT=1: the old form.
  static const bool out_on_chain_split = getenv("LYRA_HIP_RS_OUT_ON_CHAIN_SPLIT") != nullptr;
  const bool rs_out_off_chain = rs && !c->serial && (c->nsub == 1 || !out_on_chain_split);
  struct LocalScope { lyra_hip_ctx* c; ~LocalScope() { c->chunk_local = false; c->ids_stable = false; } } local_scope{c};
  c->chunk_local = c->nsub > 1 && !c->serial && enc && dec && !feats && !rs && !(F & (LYRA_HIP_STEP_DTX | LYRA_HIP_STEP_DECODER_NOISE)) &&
                   !getenv("LYRA_HIP_NO_CHUNK_LOCAL");
  for (int i = 0; i < S->n_steps; ++i) {
    const long step = S->first_step + i;
    const int set = (int)(step & 1);
    c->ids_stable = i > 0;   // one id list for every step of this call
    if (enc) {
      const int16_t* in = S->d_pcm_ring + (size_t)(step % S->ring) * B * (size_t)n_ext;
      if (rs && !rs_off_chain) {   // lyra_encoder.cc:119-122: external rate -> 16 kHz, the encoder's own resampler
        if ((rc = lyra_hip_resample_dev(c, LYRA_HIP_SIDE_ENCODER, S->d_stream_ids, S->B, in, n_ext, ext, 16000, c->d_pcm_in))) return rc;
        in = c->d_pcm_in;
        {   // the chunks of a split encode run on se[1..]: they read what se[0] has just written
          DEVSCOPE(c);
          if (c->nsub > 1) {
            HIPCHK(c, hipEventRecord(c->ev_ahead_order, c->se[0]));
            for (int k = 1; k < c->nsub; ++k) HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_ahead_order, 0));
          }
        }
      } else if (rs) {         // ... two steps ahead, on the quantizer stream (resample_in_ahead)
        DEVSCOPE(c);
        static const int lead = getenv("LYRA_HIP_RS_LEAD") ? std::max(1, std::min(2, atoi(getenv("LYRA_HIP_RS_LEAD")))) : 2;   // experiment hook
        auto ahead = [&](int j) {   // the hop of step first_step + j, if this call has one
          if (j >= S->n_steps) return 0;
          const int16_t* src = S->d_pcm_ring + (size_t)((S->first_step + j) % S->ring) * B * (size_t)n_ext;
          return resample_in_ahead(c, S->d_stream_ids, S->B, src, n_ext, ext, S->first_step + j);
        };
        if (i == 0) {
          if ((rc = ahead_begin(c))) return rc;
          for (int j = 0; j < lead; ++j) if ((rc = ahead(j))) return rc;
        }
        if ((rc = ahead(i + lead))) return rc;
        const int slot = (int)(step % lyra_hip_ctx::RS_RING);
        for (int k = 0; k < c->nsub; ++k) HIPCHK(c, hipStreamWaitEvent(c->se[k], c->ev_rs_in[slot], 0));
        in = c->d_rs16[slot];
      }
      if (F & LYRA_HIP_STEP_DTX)
        rc = lyra_hip_encode_dtx_dev(c, S->d_stream_ids, S->B, in, S->num_bits, S->d_packets[set], S->d_packet_bytes[set]);
      else
        rc = lyra_hip_encode_dev(c, S->d_stream_ids, S->B, in, S->num_bits, S->d_packets[set]);
      if (rc) return rc;
    }
    if (dec) {
      if (feats) rc = lyra_hip_generate_dev(c, S->d_stream_ids, S->B, S->d_features + (size_t)(step % (S->n_features > 0 ? S->n_features : 1)) * B * 64, S->d_pcm_out[set]);
      else {
        const uint8_t* pk = S->d_packets[set];
        if (!enc && S->d_packet_ring && S->n_packet_ring > 0)
          pk = S->d_packet_ring + (size_t)(step % S->n_packet_ring) * B * (size_t)((S->num_bits + 7) / 8);
        rc = lyra_hip_decode_dev(c, S->d_stream_ids, S->B, pk, S->num_bits, S->d_pcm_out[set]);
      }
      if (rc) return rc;
      static const bool split_sn = getenv("LYRA_HIP_SPLIT_SN_CALLS") != nullptr;   // experiment hook: the form before round 6
      if ((F & LYRA_HIP_STEP_DECODER_NOISE) && rs_out_off_chain && !split_sn) {   // both legs: one noise-stream call
        if ((rc = noise_and_resample_deferred(c, S->d_stream_ids, S->B, S->d_pcm_out[set], S->d_is_noise, ext, S->d_ext_out[set]))) return rc;
      } else {
        if (F & LYRA_HIP_STEP_DECODER_NOISE)   // lyra_decoder.cc:304-311: every decoded hop of a received packet
          if ((rc = lyra_hip_noise_receive_dev(c, LYRA_HIP_SIDE_DECODER, S->d_stream_ids, S->B, S->d_pcm_out[set], S->d_is_noise))) return rc;
        if (rs)     // lyra_decoder.cc:107-113 / buffered_resampler.cc: 16 kHz -> external rate
          if ((rc = rs_out_off_chain ? resample_deferred(c, S->d_stream_ids, S->B, S->d_pcm_out[set], 320, 16000, ext, S->d_ext_out[set])
                                 : lyra_hip_resample_dev(c, LYRA_HIP_SIDE_DECODER, S->d_stream_ids, S->B, S->d_pcm_out[set], 320, 16000,
                                                         ext, S->d_ext_out[set])))
            return rc;
      }
    }
  }
  return 0;
}

int lyra_hip_profile_enable(lyra_hip_ctx* c, unsigned kernel_mask) {
  if (!c) return LYRA_HIP_EINVAL;
  c->profiling = kernel_mask;
  for (long& n : c->prof_seen) n = 0;
  return 0;
}
int lyra_hip_profile_sample(lyra_hip_ctx* c, int every) {
  if (!c || every < 1) return LYRA_HIP_EINVAL;
  c->prof_every = every;
  return 0;
}
int lyra_hip_profile_kernel_count(void) { return K_COUNT; }
const char* lyra_hip_profile_kernel_name(int i) { return (i >= 0 && i < K_COUNT) ? kKernelNames[i] : ""; }
int lyra_hip_profile_read(lyra_hip_ctx* c, double* total_ms, long* launches) {
  if (!c || !total_ms || !launches) return LYRA_HIP_EINVAL;
  int rc = sync_all(c);
  if (rc) return rc;
  for (int i = 0; i < K_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
  for (auto& sp : c->spans) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, sp.a, sp.b));
    total_ms[sp.kid] += ms;
    launches[sp.kid] += 1;
    c->event_pool.push_back(sp.a);
    c->event_pool.push_back(sp.b);
  }
  c->spans.clear();
  return 0;
}

// Start / end of every recorded span in ms relative to the FIRST span's start (HIP events compare across streams of one
// device); call before lyra_hip_profile_read, which recycles the events.  -> spans written (at most cap).
int lyra_hip_profile_timeline(lyra_hip_ctx* c, int cap, int* kernel_ids, float* start_ms, float* end_ms) {
  if (!c || !kernel_ids || !start_ms || !end_ms) return LYRA_HIP_EINVAL;
  int rc = sync_all(c);
  if (rc) return rc;
  int n = 0;
  for (auto& sp : c->spans) {
    if (n >= cap) break;
    HIPCHK(c, hipEventElapsedTime(&start_ms[n], c->spans.front().a, sp.a));
    HIPCHK(c, hipEventElapsedTime(&end_ms[n], c->spans.front().a, sp.b));
    kernel_ids[n++] = sp.kid;
  }
  return n;
}

long lyra_hip_debug_read(lyra_hip_ctx* c, int which, float* host_out, long capacity) {
  if (!c || !host_out) return LYRA_HIP_EINVAL;
  const float* src = nullptr;
  long n = 0;
  switch (which) {
    case 0: src = c->d_e0; n = (long)c->last_B_enc * 4 * 128; break;
    case 1: src = c->d_e1; n = (long)c->last_B_enc * 2 * 256; break;
    case 2: src = c->d_codes; n = (long)c->last_B_enc * 64; break;
    case 3: src = c->d_d0; n = (long)c->last_B_dec * 4 * 128; break;
    case 4: src = c->d_d1; n = (long)c->last_B_dec * 20 * 64; break;
    case 5: {   // the quantizer's screen: frame-stages / wavefront-stages that took the exact chain since creation
      if (capacity < 2) return fail(c, LYRA_HIP_EINVAL, "debug buffer needs 2 floats");
      int rc5 = sync_all(c);
      if (rc5) return rc5;
      unsigned h[4];
      HIPCHK(c, hipMemcpy(h, c->d_rvq_stats, sizeof h, hipMemcpyDeviceToHost));
      host_out[0] = (float)h[0]; host_out[1] = (float)h[1];
      return 2;
    }
    default: return fail(c, LYRA_HIP_EINVAL, "unknown debug buffer %d", which);
  }
  if (n > capacity) return fail(c, LYRA_HIP_EINVAL, "debug buffer needs %ld floats", n);
  int rc = sync_all(c);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(host_out, src, (size_t)n * 4, hipMemcpyDeviceToHost));
  return n;
}

}  // extern "C"

#include "pipe_api.inc"
#include "twin_api.inc"
