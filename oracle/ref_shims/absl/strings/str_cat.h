#ifndef REF_SHIM_ABSL_STR_CAT_H_
#define REF_SHIM_ABSL_STR_CAT_H_
#include <sstream>
#include <string>
namespace absl {
template <typename... A>
std::string StrCat(const A&... a) {
  std::ostringstream s;
  (s << ... << a);
  return s.str();
}
}  // namespace absl
#endif
