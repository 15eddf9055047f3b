#ifndef REF_SHIM_ABSL_MEMORY_H_
#define REF_SHIM_ABSL_MEMORY_H_
#include <memory>
namespace absl {
template <typename T>
std::unique_ptr<T> WrapUnique(T* p) { return std::unique_ptr<T>(p); }
using std::make_unique;
}  // namespace absl
#endif
