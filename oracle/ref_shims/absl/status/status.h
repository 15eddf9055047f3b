#ifndef REF_SHIM_ABSL_STATUS_H_
#define REF_SHIM_ABSL_STATUS_H_
#include <ostream>
#include <string>
namespace absl {
class Status {
 public:
  Status() : ok_(true) {}
  Status(bool ok, std::string msg) : ok_(ok), msg_(std::move(msg)) {}
  bool ok() const { return ok_; }
  const std::string& message() const { return msg_; }
 private:
  bool ok_;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(const std::string& m) { return Status(false, "INVALID_ARGUMENT: " + m); }
inline Status UnknownError(const std::string& m) { return Status(false, "UNKNOWN: " + m); }
inline std::ostream& operator<<(std::ostream& o, const Status& s) { return o << (s.ok() ? std::string("OK") : s.message()); }
}  // namespace absl
#endif
