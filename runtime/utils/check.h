// Error convention of the reference runtime (runtime/core/utils/log.h:51-79): a failed CHECK prints and exit(-1)s,
// LOG(FATAL) aborts.  Minimal stand-ins with the same observable behaviour.
#ifndef RUNTIME_UTILS_CHECK_H_
#define RUNTIME_UTILS_CHECK_H_
#include <cstdio>
#include <cstdlib>
#include <sstream>

namespace wekws {
class FatalMessage {
 public:
  FatalMessage(const char* file, int line, bool is_check) : is_check_(is_check) { s_ << file << ":" << line << " "; }
  [[noreturn]] ~FatalMessage() {
    std::fprintf(stderr, "%s\n", s_.str().c_str());
    if (is_check_) std::exit(-1);
    std::abort();
  }
  std::ostream& stream() { return s_; }
 private:
  std::ostringstream s_;
  bool is_check_;
};
}  // namespace wekws

#define WEKWS_CHECK(cond) \
  if (!(cond)) ::wekws::FatalMessage(__FILE__, __LINE__, true).stream() << "Check failed: " #cond " "
#define WEKWS_FATAL() ::wekws::FatalMessage(__FILE__, __LINE__, false).stream()
#endif  // RUNTIME_UTILS_CHECK_H_
