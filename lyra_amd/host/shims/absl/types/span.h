// Minimal stand-in for absl::Span (the subset the Lyra plugin interfaces use); abseil is not available offline.
#ifndef LYRA_AMD_SHIM_ABSL_SPAN_H_
#define LYRA_AMD_SHIM_ABSL_SPAN_H_
#include <cstddef>
#include <type_traits>
#include <vector>
namespace absl {
template <typename T>
class Span {
 public:
  using value_type = std::remove_cv_t<T>;
  constexpr Span() : p_(nullptr), n_(0) {}
  constexpr Span(T* p, size_t n) : p_(p), n_(n) {}
  template <typename V, typename = std::enable_if_t<std::is_same<std::remove_cv_t<typename V::value_type>, value_type>::value>>
  Span(V& v) : p_(v.data()), n_(v.size()) {}  // NOLINT: implicit like absl
  template <typename V, typename = std::enable_if_t<std::is_const<T>::value &&
                                                    std::is_same<typename V::value_type, value_type>::value>>
  Span(const V& v) : p_(v.data()), n_(v.size()) {}  // NOLINT
  constexpr T* data() const { return p_; }
  constexpr size_t size() const { return n_; }
  constexpr size_t length() const { return n_; }
  constexpr bool empty() const { return n_ == 0; }
  constexpr T* begin() const { return p_; }
  constexpr T* end() const { return p_ + n_; }
  constexpr T& operator[](size_t i) const { return p_[i]; }
  constexpr T& at(size_t i) const { return p_[i]; }
  constexpr Span subspan(size_t pos, size_t len) const { return Span(p_ + pos, len); }
  constexpr Span first(size_t len) const { return Span(p_, len); }
  constexpr Span last(size_t len) const { return Span(p_ + n_ - len, len); }
  constexpr T& front() const { return p_[0]; }
  constexpr T& back() const { return p_[n_ - 1]; }
 private:
  T* p_;
  size_t n_;
};
template <typename T>
constexpr Span<const T> MakeConstSpan(const T* p, size_t n) { return Span<const T>(p, n); }
template <typename V>
Span<const typename V::value_type> MakeConstSpan(const V& v) { return Span<const typename V::value_type>(v.data(), v.size()); }
}  // namespace absl
#endif
