// lyra_hip_components.cc -- see lyra_hip_components.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_hip_components.h"

#include <atomic>
#include <bitset>
#include <condition_variable>
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
// the GPU form the next batch (group commit -- no timer, no added latency for a lone caller).  Results are per stream
// and bit-identical to the B = 1 calls (the kernels are batch-invariant, tests/test_gpu_parity.py).
struct CombinerStats { std::atomic<long> calls{0}, batches{0}, largest{0}; };
template <class Req>
class Combiner {
 public:
  template <class Exec>   // exec(std::vector<Req*>&): sets every request's rc
  void Run(Req* r, Exec exec) {
    std::unique_lock<std::mutex> l(mu_);
    pending_.push_back(r);
    while (!r->done) {
      if (busy_) { cv_.wait(l); continue; }
      busy_ = true;
      std::vector<Req*> batch;
      batch.swap(pending_);
      l.unlock();
      exec(batch);
      stats.calls += (long)batch.size();
      stats.batches += 1;
      long big = stats.largest.load();
      while ((long)batch.size() > big && !stats.largest.compare_exchange_weak(big, (long)batch.size())) {}
      l.lock();
      for (Req* q : batch) q->done = true;
      busy_ = false;
      cv_.notify_all();
    }
  }
  CombinerStats stats;
 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Req*> pending_;
  bool busy_ = false;
};
struct HopReq {   // one hop in, one vector out, of one stream
  int32_t id; const void* in; void* out; int arg; int rc; bool done;
};

// One GPU context per process, shared by all plugin objects; each object owns a stream id.
class SharedContext {
 public:
  static SharedContext& Get() { static SharedContext s; return s; }
  int device = 0;
  int max_streams = 1024;

  lyra_hip_ctx* Acquire(const std::string& model_dir, int* stream_id) {
    std::lock_guard<std::mutex> l(mu_);
    if (!ctx_) {
      if (lyra_hip_create(model_dir.c_str(), device, max_streams, LYRA_HIP_REQUANT_DEFAULT, &ctx_) != 0) {
        LOG(ERROR) << "lyra_hip_create failed: " << lyra_hip_last_error(nullptr);
        ctx_ = nullptr;
        return nullptr;
      }
      for (int i = max_streams - 1; i >= 0; --i) free_.push_back(i);
    }
    if (free_.empty()) { LOG(ERROR) << "No free stream slot (SetMaxStreams)."; return nullptr; }
    const int32_t id = free_.back();
    {
      // The C ABI wants every call on a context serialised, and the reset touches the same staging buffers
      // and id-stamp table as Extract / Generate running on other threads' objects: take the call mutex.
      std::lock_guard<std::mutex> lc(call_mu_);
      if (lyra_hip_reset_streams(ctx_, &id, 1) != 0) {
        LOG(ERROR) << "lyra_hip_reset_streams failed: " << lyra_hip_last_error(ctx_);
        if (users_ == 0) { lyra_hip_destroy(ctx_); ctx_ = nullptr; free_.clear(); }
        return nullptr;   // the slot stays on the free list
      }
    }
    free_.pop_back();
    ++users_;
    *stream_id = id;
    return ctx_;
  }
  void Release(int stream_id) {
    std::lock_guard<std::mutex> l(mu_);
    free_.push_back(stream_id);
    if (--users_ == 0) {
      std::lock_guard<std::mutex> lc(call_mu_);   // no call of another thread may still be inside the context
      lyra_hip_destroy(ctx_);
      ctx_ = nullptr;
      free_.clear();
    }
  }
  std::mutex& call_mutex() { return call_mu_; }  // the C ABI wants calls on one context serialised

  // ---- combined calls (see Combiner) ---------------------------------------------------------------------
  enum Kind { kExtract, kLogMel, kGenerate, kQuantize, kDequantize, kKinds };
  // in / out element counts and sizes per request of each kind
  int Call(Kind kind, lyra_hip_ctx* ctx, int32_t id, const void* in, void* out, int arg = 0) {
    HopReq r{id, in, out, arg, -1, false};
    comb_[kind].Run(&r, [&](std::vector<HopReq*>& batch) { Execute(kind, ctx, batch); });
    return r.rc;
  }
  const CombinerStats& stats(Kind kind) const { return comb_[kind].stats; }

 private:
  void Execute(Kind kind, lyra_hip_ctx* ctx, std::vector<HopReq*>& batch) {
    if (kind == kQuantize) {   // requests of different bit rates cannot share a call: one call per rate present
      std::vector<HopReq*> todo(batch), same, rest;   // (the caller still needs `batch` to mark its requests done)
      while (!todo.empty()) {
        same.clear();
        rest.clear();
        for (HopReq* q : todo) (q->arg == todo[0]->arg ? same : rest).push_back(q);
        ExecuteUniform(kind, ctx, same);
        todo.swap(rest);
      }
      return;
    }
    ExecuteUniform(kind, ctx, batch);
  }
  void ExecuteUniform(Kind kind, lyra_hip_ctx* ctx, std::vector<HopReq*>& batch) {
    static const size_t kIn[kKinds] = {kHop * sizeof(int16_t), kHop * sizeof(int16_t), kNumFeatures * sizeof(float),
                                       kNumFeatures * sizeof(float), LYRA_HIP_MAX_STAGES * sizeof(int32_t)};
    static const size_t kOut[kKinds] = {kNumFeatures * sizeof(float), LYRA_HIP_NUM_MEL * sizeof(float),
                                        kHop * sizeof(int16_t), LYRA_HIP_MAX_STAGES * sizeof(int32_t),
                                        kNumFeatures * sizeof(float)};
    const int B = (int)batch.size();
    int rc;
    std::lock_guard<std::mutex> l(call_mu_);
    if (B == 1) {   // the common uncontended case: no staging copies
      HopReq* q = batch[0];
      rc = Dispatch(kind, ctx, &q->id, 1, q->in, q->out, q->arg);
    } else {
      ids_.resize(B);
      in_.resize((size_t)B * kIn[kind]);
      out_.resize((size_t)B * kOut[kind]);
      for (int i = 0; i < B; ++i) {
        ids_[i] = batch[i]->id;
        std::memcpy(in_.data() + (size_t)i * kIn[kind], batch[i]->in, kIn[kind]);
      }
      rc = Dispatch(kind, ctx, ids_.data(), B, in_.data(), out_.data(), batch[0]->arg);
      if (rc == 0)
        for (int i = 0; i < B; ++i) std::memcpy(batch[i]->out, out_.data() + (size_t)i * kOut[kind], kOut[kind]);
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
  std::vector<int32_t> ids_;        // staging of a combined call (guarded by call_mu_)
  std::vector<uint8_t> in_, out_;
  std::mutex mu_, call_mu_;
  lyra_hip_ctx* ctx_ = nullptr;
  std::vector<int> free_;
  int users_ = 0;
};

class StreamHandle {
 public:
  explicit StreamHandle(const ghc::filesystem::path& model_path) {
    ctx_ = SharedContext::Get().Acquire(model_path.string(), &id_);
  }
  ~StreamHandle() { if (ctx_) SharedContext::Get().Release(id_); }
  bool ok() const { return ctx_ != nullptr; }
  lyra_hip_ctx* ctx() const { return ctx_; }
  int32_t id() const { return id_; }
 private:
  lyra_hip_ctx* ctx_ = nullptr;
  int id_ = -1;
};

class SoundStreamEncoderHip : public FeatureExtractorInterface {
 public:
  explicit SoundStreamEncoderHip(const ghc::filesystem::path& p) : h_(p) {}
  bool ok() const { return h_.ok(); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (static_cast<int>(audio.size()) != kHop) {
      LOG(ERROR) << "Input audio should have " << kHop << " samples but instead had " << audio.size() << ".";
      return std::nullopt;
    }
    std::vector<float> out(kNumFeatures);
    if (SharedContext::Get().Call(SharedContext::kExtract, h_.ctx(), h_.id(), audio.data(), out.data()) != 0) {
      LOG(ERROR) << "Unable to run the SoundStream encoder: " << lyra_hip_last_error(h_.ctx());
      return std::nullopt;
    }
    return out;
  }
 private:
  StreamHandle h_;
};

class LogMelHip : public FeatureExtractorInterface {
 public:
  explicit LogMelHip(const ghc::filesystem::path& p) : h_(p) {}
  bool ok() const { return h_.ok(); }
  std::optional<std::vector<float>> Extract(const absl::Span<const int16_t> audio) override {
    if (static_cast<int>(audio.size()) != kHop) {
      LOG(ERROR) << "Input audio should have " << kHop << " samples but instead had " << audio.size() << ".";
      return std::nullopt;
    }
    std::vector<float> out(LYRA_HIP_NUM_MEL);
    if (SharedContext::Get().Call(SharedContext::kLogMel, h_.ctx(), h_.id(), audio.data(), out.data()) != 0)
      return std::nullopt;
    return out;
  }
 private:
  StreamHandle h_;
};

class ResidualVectorQuantizerHip : public VectorQuantizerInterface {
 public:
  explicit ResidualVectorQuantizerHip(const ghc::filesystem::path& p) : h_(p) {}
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
    if (SharedContext::Get().Call(SharedContext::kQuantize, h_.ctx(), h_.id(), features.data(), idx, num_bits) != 0) {
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
    if (SharedContext::Get().Call(SharedContext::kDequantize, h_.ctx(), h_.id(), idx, out.data()) != 0) {
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
  LyraGanModelHip(const ghc::filesystem::path& p, int num_features) : GenerativeModel(kHop, num_features), h_(p) {}
  bool ok() const { return h_.ok(); }
 protected:
  bool RunConditioning(const std::vector<float>& features) override {
    hop_.resize(kHop);
    if (static_cast<int>(features.size()) != kNumFeatures) return false;
    return SharedContext::Get().Call(SharedContext::kGenerate, h_.ctx(), h_.id(), features.data(), hop_.data()) == 0;
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
    if (c.largest.load() > st.largest_batch) st.largest_batch = c.largest.load();
  }
  return st;
}
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
