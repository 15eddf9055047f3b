#ifndef REF_SHIM_ABSL_STATUSOR_H_
#define REF_SHIM_ABSL_STATUSOR_H_
#include <optional>
#include <utility>
#include "absl/status/status.h"
namespace absl {
template <typename T>
class StatusOr {
 public:
  StatusOr(const Status& s) : status_(s) {}                 // NOLINT: implicit like absl
  StatusOr(T&& v) { value_.emplace(std::move(v)); }          // NOLINT
  StatusOr(const T& v) { value_.emplace(v); }                // NOLINT
  bool ok() const { return value_.has_value(); }
  const Status& status() const { return status_; }
  T& value() { return *value_; }
  const T& value() const { return *value_; }
  T& operator*() { return *value_; }
  const T& operator*() const { return *value_; }
  T* operator->() { return &*value_; }
  const T* operator->() const { return &*value_; }
 private:
  Status status_;
  std::optional<T> value_;
};
}  // namespace absl
#endif
