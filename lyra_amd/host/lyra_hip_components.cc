// lyra_hip_components.cc -- see lyra_hip_components.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_hip_components.h"

#include <algorithm>
#include <atomic>
#include <bitset>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <sched.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lyra_hip.h"

namespace chromemedia {
namespace codec {
namespace {

constexpr int kNumFeatures = LYRA_HIP_NUM_FEATURES;
constexpr int kHop = LYRA_HIP_HOP;
constexpr int kMaxBits = 4 * LYRA_HIP_MAX_STAGES;

// Call combining.  Every plugin object is one stream, and the reference calls its plugins one 20 ms hop at a time
// (lyra_encoder.cc:143-155, lyra_decoder.cc:198-207): with many codec objects on many threads that is many B = 1 device
// calls queueing on the context's call mutex.  A Combiner turns whatever calls of one kind are waiting into ONE batched
// call of the C ABI: a caller appends its request; if nobody is executing it becomes the leader, takes everything that
// is pending (its own request included), runs the batch and wakes the others; requests that arrive while a batch is on
// the GPU form the next batch (group commit).  Results are per stream and bit-identical to the B = 1 calls (the kernels
// are batch-invariant, tests/test_gpu_parity.py).
//
// Round 5 (the per-object path ran at 27-33 k hops/s on 64-1,024 threads, below the 42 k of the CPU oracle on the same
// cores: profiles/history/r04_plugin_mt_throughput.txt -- batches of 10-60 requests and a condition variable that woke EVERY
// waiter of a kind after every batch):
//  * every thread sleeps on a Waiter of its own and is woken exactly once, when its own request is done (or when it is
//    handed the leadership); a finished batch is woken as a tree (the leader wakes the first kFan members, member i wakes
//    members kFan * i + 1 ... kFan * i + kFan), so the wake-ups of a 1,000-request batch run on all host cores instead
//    of serially on the leader;
//  * leadership is handed to the oldest pending request before the finished batch is woken: the next device call starts
//    while the previous batch's callers are still being scheduled;
//  * adaptive gathering: callers that drive their codecs in lock-step (N threads, one hop each per round) arrive over
//    the time the scheduler needs to wake them.  A leader that finds fewer requests pending than the previous two batches
//    of this kind held takes the stragglers along: it polls (sched_yield: the arriving threads get the core) until that
//    many are pending, or nothing has arrived for kQuietUs, or kGatherBaseUs + kGatherPerReqUs per expected request
//    (capped at kGatherMaxUs) have passed.  A lone caller (previous batches of one request) never waits, so a single
//    codec's latency is unchanged (plugin_demo --bench); when callers leave, the expectation follows the batches actually
//    seen within two calls.
struct CombinerStats { std::atomic<long> calls{0}, batches{0}, largest{0}, gather_us{0}, exec_us{0}, timeouts{0}; };
struct Waiter {   // one per host thread (thread_local, shared: a waker may still hold it while the thread exits)
  std::mutex m;
  std::condition_variable cv;
};
template <class Req>
class Combiner {
 public:
  enum { kPending = 0, kDone = 1, kLead = 2 };
  // tunables (environment, read once; plugin_mt_sweep.sh (a probe of an earlier round, removed since: git history): 10-150 us of quiet time and fan-outs 2-16 are within
  // run-to-run spread of each other at 256 and 1,024 threads)
  static long EnvLong(const char* name, long dflt) { const char* v = std::getenv(name); return v ? std::atol(v) : dflt; }
  const int kFan = (int)std::max(1L, EnvLong("LYRA_HIP_COMBINER_FAN", 4));
  const long kQuietUs = EnvLong("LYRA_HIP_COMBINER_QUIET_US", 40);
  const long kGatherBaseUs = EnvLong("LYRA_HIP_COMBINER_GATHER_US", 100), kGatherPerReqUs = 2, kGatherMaxUs = 3000;
  // a gathering leader polls (sched_yield) for this long only, then sleeps on its Waiter until the arrival that completes
  // the batch wakes it, the quiet time passes or the budget is spent (round 6: up to five leaders used to spin for up to
  // 3 ms each -- in a CPU-quota-limited container that is quota the arriving threads need, and under SCHED_FIFO
  // sched_yield() does not give the core to a lower-priority arrival at all)
  // 300 us measured as the knee (profiles/r06_plugin_mt_spin.txt, 64 / 256 / 1,024 threads): 30 us 71 / 173 / 303 k frames/s,
  // 300 us 77 / 199 / 316 k, 3,000 us (= no sleeping, the round-5 behaviour) 78 / 197 / 308 k.
  const long kSpinUs = EnvLong("LYRA_HIP_COMBINER_SPIN_US", 300);
  template <class Exec>   // exec(std::vector<Req*>&): sets every request's rc
  void Run(Req* r, Exec exec) {
    thread_local std::shared_ptr<Waiter> me = std::make_shared<Waiter>();
    r->waiter = me;
    r->state = kPending;
    std::unique_lock<std::mutex> l(mu_);
    pending_.push_back(r);
    npending_.store((long)pending_.size(), std::memory_order_release);
    if (busy_) {
      // a leader asleep in its gathering phase is woken by the request that completes the batch it expects
      std::shared_ptr<Waiter> sleeper = (gather_waiter_ && (long)pending_.size() >= gather_expect_) ? gather_waiter_ : nullptr;
      l.unlock();
      if (sleeper) { { std::lock_guard<std::mutex> lw(sleeper->m); } sleeper->cv.notify_one(); }
      {
        std::unique_lock<std::mutex> lw(me->m);
        me->cv.wait(lw, [&] { return r->state != kPending; });
      }
      if (r->state == kDone) {   // woken by the batch's leader or by the member above this one in the wake tree
        WakeChildren(r);
        return;
      }
      l.lock();                  // kLead: the previous leader handed the (still busy) combiner to this request
    } else {
      busy_ = true;
    }
    // ---- leader ----
    const long expect = std::max(last_[0], last_[1]);
    if (expect > 1 && (long)pending_.size() < expect) {
      // gather: requests are still arriving (the callers of the previous batch are being scheduled); take them along --
      // until as many as before are pending, or nothing has arrived for kQuietUs, or the budget is spent.  Polled with
      // sched_yield(): the threads that are about to arrive get this core.
      gather_waiter_ = me;
      gather_expect_ = expect;
      l.unlock();
      const auto g0 = std::chrono::steady_clock::now();
      const auto deadline = g0 + std::chrono::microseconds(std::min(kGatherMaxUs, kGatherBaseUs + kGatherPerReqUs * expect));
      const auto quiet = std::chrono::microseconds(kQuietUs), spin = std::chrono::microseconds(kSpinUs);
      auto last_arrival = g0;
      long seen = npending_.load(std::memory_order_acquire);
      for (;;) {
        const auto now = std::chrono::steady_clock::now();
        const long have = npending_.load(std::memory_order_acquire);
        if (have != seen) { seen = have; last_arrival = now; }
        if (have >= expect || now >= deadline || now - last_arrival >= quiet) break;
        if (now - g0 < spin) { sched_yield(); continue; }
        // asleep until the batch is complete (notified), the quiet time has passed without an arrival, or the deadline
        std::unique_lock<std::mutex> lw(me->m);
        me->cv.wait_for(lw, std::min<std::chrono::steady_clock::duration>(deadline - now, quiet - (now - last_arrival)),
                        [&] { return npending_.load(std::memory_order_acquire) >= expect; });
      }
      stats.gather_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g0).count();
      if (seen < expect) stats.timeouts += 1;
      l.lock();
      gather_waiter_ = nullptr;
    }
    auto batch = std::make_shared<std::vector<Req*>>();
    batch->swap(pending_);
    npending_.store(0, std::memory_order_relaxed);
    l.unlock();
    const auto e0 = std::chrono::steady_clock::now();
    exec(*batch);
    stats.exec_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - e0).count();
    const long n = (long)batch->size();
    stats.calls += n;
    stats.batches += 1;
    long big = stats.largest.load();
    while (n > big && !stats.largest.compare_exchange_weak(big, n)) {}
    Req* next = nullptr;
    l.lock();
    last_[1] = last_[0];
    last_[0] = n;
    if (pending_.empty()) busy_ = false; else next = pending_.front();   // else: busy_ stays set for `next`
    l.unlock();
    if (next) Signal(next, kLead);
    // wake the batch: this request's slot in the tree is taken by the batch's first other member
    std::vector<Req*>& b = *batch;
    int self = 0;
    for (int i = 0; i < (int)b.size(); ++i) if (b[i] == r) { self = i; break; }
    std::swap(b[0], b[self]);   // the leader is the root (index 0) and needs no wake-up
    for (int i = 0; i < (int)b.size(); ++i) { b[i]->tree = batch; b[i]->tree_index = i; }
    WakeChildren(r);
  }
  CombinerStats stats;
 private:
  static void Signal(Req* q, int state) {
    std::shared_ptr<Waiter> w = q->waiter;   // q lives on its caller's stack and may be gone right after the store
    { std::lock_guard<std::mutex> lw(w->m); q->state = state; }
    w->cv.notify_one();
  }
  // member i of a finished batch wakes members kFan * i + 1 ... kFan * i + kFan (all still blocked, so their requests
  // are alive; the vector itself is shared, the leader may have returned)
  void WakeChildren(Req* r) {
    const std::shared_ptr<std::vector<Req*>> tree = std::move(r->tree);
    if (!tree) return;
    const long n = (long)tree->size(), i = r->tree_index;
    for (long c = kFan * i + 1; c <= kFan * i + kFan && c < n; ++c) Signal((*tree)[c], kDone);
  }
  std::mutex mu_;
  std::vector<Req*> pending_;
  std::atomic<long> npending_{0};       // pending_.size(), readable without mu_ by the gathering leader
  bool busy_ = false;
  long last_[2] = {1, 1};               // sizes of the two latest batches
  std::shared_ptr<Waiter> gather_waiter_;   // the leader that is gathering right now (guarded by mu_), and what it waits for
  long gather_expect_ = 0;
};
struct HopReq {   // one hop in, one vector out, of one stream
  int32_t id; const void* in; void* out; int arg; int rc;
  int state = 0;                                   // Combiner: kPending / kDone / kLead, guarded by waiter->m
  std::shared_ptr<Waiter> waiter;
  std::shared_ptr<std::vector<HopReq*>> tree;      // the finished batch this request belongs to, and its place in it
  long tree_index = 0;
};

// Two GPU contexts per process, shared by all plugin objects; each object owns a stream id (valid in both).
// The C ABI wants the calls on ONE context serialised (they share its staging buffers), and the plugin contract makes every
// hop four blocking device calls: with one context the four kinds of calls queue behind one mutex.  Encoder state and
// decoder state of a stream are disjoint (state_layout.h: one region per kernel), so the extractor-side kinds (Extract,
// log-mel, Quantize) run on one context and the decoder-side kinds (DecodeToLossyFeatures, AddFeatures + GenerateSamples)
// on a second one: two device calls in flight, on streams that the small-context CU partition already keeps on
// complementary halves of the chip.  A context costs a copy of the 3 MB of weights, the per-stream state of max_streams
// streams and its staging buffers, so each side's context is created on demand (round 6): by the first object that can only
// ever use that side (feature extractor / log-mel: extractor side; generative model: decoder side), or by the first call of
// that side (a quantizer object needs the extractor side for Quantize and the decoder side for DecodeToLossyFeatures; it
// makes sure ONE context exists when it is created, so that a bad model path still fails in the factory).  A process that
// only encodes or only decodes holds one context.
class SharedContext {
 public:
  static SharedContext& Get() { static SharedContext s; return s; }
  int device = 0;
  int max_streams = 1024;
  enum Side { kEncSide, kDecSide, kSides };

  // `needs`: the side this object will certainly use (kSides = either will do, as long as one exists).  -> success
  bool Acquire(const std::string& model_dir, int* stream_id, int needs) {
    std::lock_guard<std::mutex> l(mu_);
    if (users_ == 0) {
      model_dir_ = model_dir;
      streams_ = max_streams;
      free_.clear();
      for (int i = streams_ - 1; i >= 0; --i) free_.push_back(i);
    }
    if (needs == kSides) needs = side_[kDecSide].ctx && !side_[kEncSide].ctx ? kDecSide : kEncSide;
    {
      std::lock_guard<std::mutex> lc(side_[needs].call_mu);
      if (!EnsureContext(needs)) {
        if (users_ == 0) DestroyAll(/*locked_side=*/needs);
        return false;
      }
    }
    if (free_.empty()) { LOG(ERROR) << "No free stream slot (SetMaxStreams)."; return false; }
    const int32_t id = free_.back();
    for (int k = 0; k < kSides; ++k) {
      // The reset touches the same staging buffers and id-stamp table as the calls running on other threads' objects:
      // take the context's call mutex.  A side whose context does not exist yet needs no reset: it is created fresh.
      std::lock_guard<std::mutex> lc(side_[k].call_mu);
      if (side_[k].ctx && lyra_hip_reset_streams(side_[k].ctx, &id, 1) != 0) {
        LOG(ERROR) << "lyra_hip_reset_streams failed: " << lyra_hip_last_error(side_[k].ctx);
        if (users_ == 0) DestroyAll(k);
        return false;   // the slot stays on the free list
      }
    }
    free_.pop_back();
    ++users_;
    *stream_id = id;
    return true;
  }
  void Release(int stream_id) {
    std::lock_guard<std::mutex> l(mu_);
    free_.push_back(stream_id);
    if (--users_ == 0) DestroyAll();
  }

  // ---- combined calls (see Combiner) ---------------------------------------------------------------------
  enum Kind { kExtract, kLogMel, kGenerate, kQuantize, kDequantize, kKinds };
  static Side SideOf(Kind kind) { return kind == kGenerate || kind == kDequantize ? kDecSide : kEncSide; }
  const char* LastError(Kind kind) { return lyra_hip_last_error(side_[SideOf(kind)].ctx); }
  int contexts_alive() {
    std::lock_guard<std::mutex> l(mu_);
    return (side_[0].ctx ? 1 : 0) + (side_[1].ctx ? 1 : 0);
  }
  int Call(Kind kind, int32_t id, const void* in, void* out, int arg = 0) {
    HopReq r{id, in, out, arg, -1};
    comb_[kind].Run(&r, [&](std::vector<HopReq*>& batch) { Execute(kind, batch); });
    return r.rc;
  }
  const CombinerStats& stats(Kind kind) const { return comb_[kind].stats; }

 private:
  struct PerSide {
    lyra_hip_ctx* ctx = nullptr;
    std::mutex call_mu;               // the C ABI wants the calls on one context serialised
    std::vector<int32_t> ids;         // staging of a combined call (guarded by call_mu)
    std::vector<uint8_t> in, out;
  };
  // call_mu of the side held.  The model path and stream count are those of the first Acquire since the last DestroyAll.
  bool EnsureContext(int k) {
    if (side_[k].ctx) return true;
    if (lyra_hip_create(model_dir_.c_str(), device, streams_, LYRA_HIP_REQUANT_DEFAULT, &side_[k].ctx) != 0) {
      LOG(ERROR) << "lyra_hip_create failed: " << lyra_hip_last_error(nullptr);
      side_[k].ctx = nullptr;
      return false;
    }
    return true;
  }
  void DestroyAll(int locked_side = -1) {   // mu_ held
    for (int k = 0; k < kSides; ++k) {
      std::unique_lock<std::mutex> lc(side_[k].call_mu, std::defer_lock);   // no call of another thread may still be inside
      if (k != locked_side) lc.lock();
      if (side_[k].ctx) lyra_hip_destroy(side_[k].ctx);
      side_[k].ctx = nullptr;
    }
    free_.clear();
  }
  void Execute(Kind kind, std::vector<HopReq*>& batch) {
    if (kind == kQuantize) {   // requests of different bit rates cannot share a call: one call per rate present
      std::vector<HopReq*> todo(batch), same, rest;   // (the caller still needs `batch` to mark its requests done)
      while (!todo.empty()) {
        same.clear();
        rest.clear();
        for (HopReq* q : todo) (q->arg == todo[0]->arg ? same : rest).push_back(q);
        ExecuteUniform(kind, same);
        todo.swap(rest);
      }
      return;
    }
    ExecuteUniform(kind, batch);
  }
  void ExecuteUniform(Kind kind, std::vector<HopReq*>& batch) {
    static const size_t kIn[kKinds] = {kHop * sizeof(int16_t), kHop * sizeof(int16_t), kNumFeatures * sizeof(float),
                                       kNumFeatures * sizeof(float), LYRA_HIP_MAX_STAGES * sizeof(int32_t)};
    static const size_t kOut[kKinds] = {kNumFeatures * sizeof(float), LYRA_HIP_NUM_MEL * sizeof(float),
                                        kHop * sizeof(int16_t), LYRA_HIP_MAX_STAGES * sizeof(int32_t),
                                        kNumFeatures * sizeof(float)};
    const int B = (int)batch.size();
    int rc;
    PerSide& S = side_[SideOf(kind)];
    std::lock_guard<std::mutex> l(S.call_mu);
    if (!EnsureContext(SideOf(kind))) {   // first call of this side (a quantizer object's other half): created fresh
      for (HopReq* q : batch) q->rc = LYRA_HIP_EHIP;
      return;
    }
    if (B == 1) {   // the common uncontended case: no staging copies
      HopReq* q = batch[0];
      rc = Dispatch(kind, S.ctx, &q->id, 1, q->in, q->out, q->arg);
    } else {
      S.ids.resize(B);
      S.in.resize((size_t)B * kIn[kind]);
      S.out.resize((size_t)B * kOut[kind]);
      for (int i = 0; i < B; ++i) {
        S.ids[i] = batch[i]->id;
        std::memcpy(S.in.data() + (size_t)i * kIn[kind], batch[i]->in, kIn[kind]);
      }
      rc = Dispatch(kind, S.ctx, S.ids.data(), B, S.in.data(), S.out.data(), batch[0]->arg);
      if (rc == 0)
        for (int i = 0; i < B; ++i) std::memcpy(batch[i]->out, S.out.data() + (size_t)i * kOut[kind], kOut[kind]);
    }
    for (HopReq* q : batch) q->rc = rc;
  }
  static int Dispatch(Kind kind, lyra_hip_ctx* ctx, const int32_t* ids, int B, const void* in, void* out, int arg) {
    switch (kind) {
      case kExtract: return lyra_hip_extract(ctx, ids, B, static_cast<const int16_t*>(in), static_cast<float*>(out));
      case kLogMel: return lyra_hip_logmel(ctx, ids, B, static_cast<const int16_t*>(in), static_cast<float*>(out));
      case kGenerate: return lyra_hip_generate(ctx, ids, B, static_cast<const float*>(in), static_cast<int16_t*>(out));
      case kQuantize: return lyra_hip_rvq_encode(ctx, B, static_cast<const float*>(in), arg, static_cast<int32_t*>(out));
      case kDequantize: return lyra_hip_rvq_decode(ctx, B, static_cast<const int32_t*>(in), static_cast<float*>(out));
      default: return LYRA_HIP_EINVAL;
    }
  }
  Combiner<HopReq> comb_[kKinds];
  PerSide side_[kSides];
  std::mutex mu_;
  std::vector<int> free_;
  int users_ = 0;
  std::string model_dir_;   // of the objects alive (the first Acquire since the last DestroyAll)
  int streams_ = 0;
};

class StreamHandle {
 public:
  StreamHandle(const ghc::filesystem::path& model_path, int needs_side) {
    ok_ = SharedContext::Get().Acquire(model_path.string(), &id_, needs_side);
  }
  ~StreamHandle() { if (ok_) SharedContext::Get().Release(id_); }
  bool ok() const { return ok_; }
  int32_t id() const { return id_; }
 private:
  bool ok_ = false;
  int id_ = -1;
};

class SoundStreamEncoderHip : public FeatureExtractorInterface {
 public:
  explicit SoundStreamEncoderHip(const ghc::filesystem::path& p) : h_(p, SharedContext::kEncSide) {}
  bool ok() const { return h_.ok(); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (static_cast<int>(audio.size()) != kHop) {
      LOG(ERROR) << "Input audio should have " << kHop << " samples but instead had " << audio.size() << ".";
      return std::nullopt;
    }
    std::vector<float> out(kNumFeatures);
    if (SharedContext::Get().Call(SharedContext::kExtract, h_.id(), audio.data(), out.data()) != 0) {
      LOG(ERROR) << "Unable to run the SoundStream encoder: " << SharedContext::Get().LastError(SharedContext::kExtract);
      return std::nullopt;
    }
    return out;
  }
 private:
  StreamHandle h_;
};

class LogMelHip : public FeatureExtractorInterface {
 public:
  explicit LogMelHip(const ghc::filesystem::path& p) : h_(p, SharedContext::kEncSide) {}
  bool ok() const { return h_.ok(); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (static_cast<int>(audio.size()) != kHop) {
      LOG(ERROR) << "Input audio should have " << kHop << " samples but instead had " << audio.size() << ".";
      return std::nullopt;
    }
    std::vector<float> out(LYRA_HIP_NUM_MEL);
    if (SharedContext::Get().Call(SharedContext::kLogMel, h_.id(), audio.data(), out.data()) != 0)
      return std::nullopt;
    return out;
  }
 private:
  StreamHandle h_;
};

class ResidualVectorQuantizerHip : public VectorQuantizerInterface {
 public:
  explicit ResidualVectorQuantizerHip(const ghc::filesystem::path& p) : h_(p, SharedContext::kSides) {}
  bool ok() const { return h_.ok(); }
  std::optional<std::string> Quantize(const std::vector<float>& features, int num_bits) const override {
    if (num_bits > kMaxBits) {
      LOG(ERROR) << "The number of bits cannot exceed maximum (" << kMaxBits << ").";
      return std::nullopt;
    }
    if (num_bits % 4 != 0 || num_bits < 0) {
      LOG(ERROR) << "The number of bits (" << num_bits << ") has to be divisible by the number of bits per quantizer (4).";
      return std::nullopt;
    }
    if (static_cast<int>(features.size()) != kNumFeatures) return std::nullopt;
    if (num_bits == 0) return std::string();
    int32_t idx[LYRA_HIP_MAX_STAGES];
    if (SharedContext::Get().Call(SharedContext::kQuantize, h_.id(), features.data(), idx, num_bits) != 0) {
      LOG(ERROR) << "Unable to invoke the quantize runner.";
      return std::nullopt;
    }
    std::string bits;
    bits.reserve(num_bits);
    for (int i = 0; i < num_bits / 4; ++i) bits += std::bitset<4>(idx[i]).to_string();  // first stage = MSBs
    return bits;
  }
  std::optional<std::vector<float>> DecodeToLossyFeatures(const std::string& quantized) const override {
    const int num_bits = static_cast<int>(quantized.size());
    if (num_bits > kMaxBits) {
      LOG(ERROR) << "The number of bits cannot exceed maximum (" << kMaxBits << ").";
      return std::nullopt;
    }
    if (num_bits % 4 != 0) {
      LOG(ERROR) << "The number of bits (" << num_bits << ") has to be divisible by the number of bits per quantizer (4).";
      return std::nullopt;
    }
    int32_t idx[LYRA_HIP_MAX_STAGES];
    for (int i = 0; i < LYRA_HIP_MAX_STAGES; ++i)
      idx[i] = i < num_bits / 4 ? static_cast<int32_t>(std::bitset<4>(quantized.substr(4 * i, 4)).to_ulong()) : -1;
    std::vector<float> out(kNumFeatures);
    if (SharedContext::Get().Call(SharedContext::kDequantize, h_.id(), idx, out.data()) != 0) {
      LOG(ERROR) << "Unable to invoke the decode runner.";
      return std::nullopt;
    }
    return out;
  }
 private:
  StreamHandle h_;
};

class LyraGanModelHip : public GenerativeModel {
 public:
  LyraGanModelHip(const ghc::filesystem::path& p, int num_features)
      : GenerativeModel(kHop, num_features), h_(p, SharedContext::kDecSide) {}
  bool ok() const { return h_.ok(); }
 protected:
  bool RunConditioning(const std::vector<float>& features) override {
    hop_.resize(kHop);
    if (static_cast<int>(features.size()) != kNumFeatures) return false;
    return SharedContext::Get().Call(SharedContext::kGenerate, h_.id(), features.data(), hop_.data()) == 0;
  }
  std::optional<std::vector<int16_t>> RunModel(int num_samples) override {
    return std::vector<int16_t>(hop_.begin() + next_sample_in_hop(), hop_.begin() + next_sample_in_hop() + num_samples);
  }
 private:
  StreamHandle h_;
  std::vector<int16_t> hop_;
};

template <class T, class... A>
std::unique_ptr<T> MakeOrNull(A&&... a) {
  auto p = std::make_unique<T>(std::forward<A>(a)...);
  if (!p->ok()) return nullptr;
  return p;
}

}  // namespace

HipCallStats GetHipCallStats() {
  HipCallStats st;
  for (int k = 0; k < SharedContext::kKinds; ++k) {
    const CombinerStats& c = SharedContext::Get().stats(static_cast<SharedContext::Kind>(k));
    st.calls += c.calls.load();
    st.device_calls += c.batches.load();
    st.gather_us += c.gather_us.load();
    st.exec_us += c.exec_us.load();
    st.gather_timeouts += c.timeouts.load();
    if (c.largest.load() > st.largest_batch) st.largest_batch = c.largest.load();
  }
  return st;
}
int GetHipContextCount() { return SharedContext::Get().contexts_alive(); }
void SetHipDevice(int device) { SharedContext::Get().device = device; }
void SetMaxStreams(int n) { SharedContext::Get().max_streams = n; }

std::unique_ptr<VectorQuantizerInterface> CreateQuantizer(const ghc::filesystem::path& model_path) {
  return MakeOrNull<ResidualVectorQuantizerHip>(model_path);
}
std::unique_ptr<GenerativeModelInterface> CreateGenerativeModel(int num_output_features,
                                                                const ghc::filesystem::path& model_path) {
  return MakeOrNull<LyraGanModelHip>(model_path, num_output_features);
}
std::unique_ptr<FeatureExtractorInterface> CreateFeatureExtractor(const ghc::filesystem::path& model_path) {
  return MakeOrNull<SoundStreamEncoderHip>(model_path);
}
std::unique_ptr<FeatureExtractorInterface> CreateLogMelExtractor(const ghc::filesystem::path& model_path) {
  return MakeOrNull<LogMelHip>(model_path);
}

}  // namespace codec
}  // namespace chromemedia
