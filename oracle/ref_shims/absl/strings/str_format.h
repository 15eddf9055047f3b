// stand-in: absl::StrFormat over snprintf.  Strings / string_views become C strings; arithmetic arguments pass through
// (the reference uses %s, %d, %f-style conversions only on this path: error messages and lyra_benchmark's table).
#ifndef REF_SHIM_ABSL_STR_FORMAT_H_
#define REF_SHIM_ABSL_STR_FORMAT_H_
#include <cstdio>
#include <sstream>
#include <string>
#include <string_view>
#include <type_traits>
namespace absl {
namespace shim_detail {
template <typename T>
auto Arg(const T& v, std::string* keep) {
  if constexpr (std::is_arithmetic_v<T> || std::is_pointer_v<T>) {
    (void)keep;
    return v;
  } else if constexpr (std::is_convertible_v<T, std::string_view>) {
    *keep = std::string(std::string_view(v));
    return keep->c_str();
  } else {
    std::ostringstream s;
    s << v;
    *keep = s.str();
    return keep->c_str();
  }
}
}  // namespace shim_detail
template <typename... A>
std::string StrFormat(std::string_view fmt, const A&... a) {
  const std::string f(fmt);
  std::string keep[sizeof...(A) + 1];
  size_t i = 0;
  auto call = [&](auto... c) {
    const int n = std::snprintf(nullptr, 0, f.c_str(), c...);
    std::string out((size_t)(n > 0 ? n : 0), '\0');
    if (n > 0) std::snprintf(out.data(), (size_t)n + 1, f.c_str(), c...);
    return out;
  };
  return call(shim_detail::Arg(a, &keep[i++])...);
}
}  // namespace absl
#endif
