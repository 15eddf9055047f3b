// lyra_hip_components.cc -- see lyra_hip_components.h.  Plain C++17 over the C ABI (include/lyra_hip.h); no HIP here.
#include "lyra_hip_components.h"

#include <bitset>
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

// One GPU context per process, shared by all plugin objects; each object owns a stream id.
class SharedContext {
 public:
  static SharedContext& Get() { static SharedContext s; return s; }
  int device = 0;
  int max_streams = 1024;

  lyra_hip_ctx* Acquire(const std::string& model_dir, int* stream_id) {
    std::lock_guard<std::mutex> l(mu_);
    if (!ctx_) {
      if (lyra_hip_create(model_dir.c_str(), device, max_streams, LYRA_HIP_REQUANT_EXACT, &ctx_) != 0) {
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

 private:
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
    int32_t id = h_.id();
    std::lock_guard<std::mutex> l(SharedContext::Get().call_mutex());
    if (lyra_hip_extract(h_.ctx(), &id, 1, audio.data(), out.data()) != 0) {
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
    int32_t id = h_.id();
    std::lock_guard<std::mutex> l(SharedContext::Get().call_mutex());
    if (lyra_hip_logmel(h_.ctx(), &id, 1, audio.data(), out.data()) != 0) return std::nullopt;
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
    {
      std::lock_guard<std::mutex> l(SharedContext::Get().call_mutex());
      if (lyra_hip_rvq_encode(h_.ctx(), 1, features.data(), num_bits, idx) != 0) {
        LOG(ERROR) << "Unable to invoke the quantize runner.";
        return std::nullopt;
      }
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
    std::lock_guard<std::mutex> l(SharedContext::Get().call_mutex());
    if (lyra_hip_rvq_decode(h_.ctx(), 1, idx, out.data()) != 0) {
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
    int32_t id = h_.id();
    std::lock_guard<std::mutex> l(SharedContext::Get().call_mutex());
    return lyra_hip_generate(h_.ctx(), &id, 1, features.data(), hop_.data()) == 0;
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
