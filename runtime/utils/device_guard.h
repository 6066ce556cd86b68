// Scoped "make this instance's GPU current": every entry point of a runtime object that allocates, frees, copies or
// launches does so under one of these, and the caller's current device is restored on the way out -- an instance
// created for device 1 works from a thread whose current device is 0 (and leaves it at 0).
#ifndef RUNTIME_UTILS_DEVICE_GUARD_H_
#define RUNTIME_UTILS_DEVICE_GUARD_H_
#include <hip/hip_runtime_api.h>

namespace wekws {
class ScopedDevice {
 public:
  explicit ScopedDevice(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    ok_ = prev_ == device || hipSetDevice(device) == hipSuccess;
    if (prev_ == device) prev_ = -1;   // nothing to restore
  }
  ~ScopedDevice() {
    if (prev_ >= 0) (void)hipSetDevice(prev_);
  }
  ScopedDevice(const ScopedDevice&) = delete;
  ScopedDevice& operator=(const ScopedDevice&) = delete;
  bool ok() const { return ok_; }

 private:
  int prev_ = -1;
  bool ok_ = false;
};
}  // namespace wekws
#endif  // RUNTIME_UTILS_DEVICE_GUARD_H_
