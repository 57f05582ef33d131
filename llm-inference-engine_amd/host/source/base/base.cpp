#include "base/base.h"

namespace base {
namespace logging {
int& min_severity() {
  static int sev = [] {
    const char* e = std::getenv("KX_LOG_LEVEL");  // 0 info .. 3 fatal
    return e ? std::atoi(e) : (int)WARNING;
  }();
  return sev;
}
}  // namespace logging
}  // namespace base
