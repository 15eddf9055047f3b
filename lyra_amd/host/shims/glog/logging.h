// Minimal stand-in for the glog macros the Lyra interfaces use (LOG(sev) << ..., VLOG, CHECK*).
#ifndef LYRA_AMD_SHIM_GLOG_H_
#define LYRA_AMD_SHIM_GLOG_H_
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace lyra_shim {
class LogLine {
 public:
  LogLine(const char* sev, bool fatal, bool on) : fatal_(fatal), on_(on) { if (on_) s_ << "[" << sev << "] "; }
  ~LogLine() { if (on_) std::cerr << s_.str() << std::endl; if (fatal_) std::abort(); }
  template <typename T> LogLine& operator<<(const T& v) { if (on_) s_ << v; return *this; }
 private:
  std::ostringstream s_;
  bool fatal_, on_;
};
}  // namespace lyra_shim
#define LOG(sev) ::lyra_shim::LogLine(#sev, false, true)
#define VLOG(n) ::lyra_shim::LogLine("V", false, false)
#define CHECK(c) if (!(c)) ::lyra_shim::LogLine("FATAL", true, true) << "Check failed: " #c " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#endif
