#ifndef REF_SHIM_ABSL_STRING_VIEW_H_
#define REF_SHIM_ABSL_STRING_VIEW_H_
#include <string_view>
namespace absl { using string_view = std::string_view; }
#endif
