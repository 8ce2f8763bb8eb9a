// Host-side helpers shared by the three C-ABI translation units (at3hip.hip, at1hip.hip, at3phip.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace at3host {

// Every entry point works on the device its context was created on, whatever device the calling thread has current
// (torch, another context on another GPU ...), and leaves the caller's current device as it found it.
class DeviceGuard {
public:
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        err_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
        changed_ = (err_ == hipSuccess && prev_ != device);
    }
    ~DeviceGuard()
    {
        if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    hipError_t error() const { return err_; }

private:
    int prev_ = -1;
    bool changed_ = false;
    hipError_t err_ = hipSuccess;
};

constexpr int kMaxGridY = 65535;   // gridDim.y / gridDim.z limit of the HIP launch interface

}  // namespace at3host
