#ifndef REF_SHIM_ABSL_MARSHALLING_H_
#define REF_SHIM_ABSL_MARSHALLING_H_
#include <string>
#include <string_view>
#include <vector>
namespace absl {
// the one overload decoder_main_lib.cc uses: a comma-separated list
inline bool ParseFlag(std::string_view text, std::vector<std::string>* out, std::string*) {
  out->clear();
  if (text.empty()) return true;
  size_t a = 0;
  while (true) {
    const size_t b = text.find(',', a);
    out->emplace_back(text.substr(a, b == std::string_view::npos ? b : b - a));
    if (b == std::string_view::npos) break;
    a = b + 1;
  }
  return true;
}
}  // namespace absl
#endif
