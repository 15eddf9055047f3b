// stand-in for absl::BitGen / BitGenRef / Uniform as decoder_main_lib.cc uses them (random request sizes)
#ifndef REF_SHIM_ABSL_RANDOM_H_
#define REF_SHIM_ABSL_RANDOM_H_
#include <random>
namespace absl {
class BitGen {
 public:
  std::mt19937_64& engine() { return e_; }
 private:
  std::mt19937_64 e_{0x4C797261ull};
};
class BitGenRef {
 public:
  BitGenRef(BitGen& g) : g_(&g) {}   // NOLINT: implicit like absl
  std::mt19937_64& engine() { return g_->engine(); }
 private:
  BitGen* g_;
};
struct IntervalOpenClosedTag {};
inline constexpr IntervalOpenClosedTag IntervalOpenClosed{};
template <typename T>
T Uniform(IntervalOpenClosedTag, BitGenRef g, T lo, T hi) {   // (lo, hi]
  return std::uniform_int_distribution<T>(lo + 1, hi)(g.engine());
}
}  // namespace absl
#endif
