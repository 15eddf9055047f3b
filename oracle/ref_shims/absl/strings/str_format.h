// stand-in: absl::StrFormat is only used for error messages on this path -- the format string, then the arguments
#ifndef REF_SHIM_ABSL_STR_FORMAT_H_
#define REF_SHIM_ABSL_STR_FORMAT_H_
#include <sstream>
#include <string>
namespace absl {
template <typename... A>
std::string StrFormat(const char* fmt, const A&... a) {
  std::ostringstream s;
  s << fmt << " [";
  ((s << " " << a), ...);
  s << " ]";
  return s.str();
}
}  // namespace absl
#endif
