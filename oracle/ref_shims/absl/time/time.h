#ifndef REF_SHIM_ABSL_TIME_H_
#define REF_SHIM_ABSL_TIME_H_
#include <chrono>
#include <cstdint>
namespace absl {
using Duration = std::chrono::duration<double>;
using Time = std::chrono::steady_clock::time_point;
inline int64_t ToInt64Seconds(Duration d) { return (int64_t)d.count(); }
inline double ToDoubleSeconds(Duration d) { return d.count() > 0 ? d.count() : 1e-9; }
// lyra_benchmark_lib.cc only ever subtracts two of these: the epoch of the steady clock will do
inline int64_t ToUnixMicros(Time t) {
  return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(t.time_since_epoch()).count();
}
}  // namespace absl
#endif
