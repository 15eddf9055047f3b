// stand-in: absl::Substitute with $0..$9 placeholders (lyra_benchmark_lib.cc builds a file name with it)
#ifndef REF_SHIM_ABSL_SUBSTITUTE_H_
#define REF_SHIM_ABSL_SUBSTITUTE_H_
#include <sstream>
#include <string>
#include <string_view>
#include <vector>
namespace absl {
template <typename... A>
std::string Substitute(std::string_view fmt, const A&... a) {
  std::vector<std::string> args;
  (([&] { std::ostringstream s; s << a; args.push_back(s.str()); })(), ...);
  std::string out;
  for (size_t i = 0; i < fmt.size(); ++i) {
    if (fmt[i] == '$' && i + 1 < fmt.size()) {
      const char c = fmt[++i];
      if (c >= '0' && c <= '9' && (size_t)(c - '0') < args.size()) out += args[(size_t)(c - '0')];
      else if (c == '$') out += '$';
    } else {
      out += fmt[i];
    }
  }
  return out;
}
}  // namespace absl
#endif
