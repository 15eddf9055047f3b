#ifndef REF_SHIM_ABSL_CLOCK_H_
#define REF_SHIM_ABSL_CLOCK_H_
#include "absl/time/time.h"
namespace absl { inline Time Now() { return std::chrono::steady_clock::now(); } }
#endif
