// host_util.hpp -- small host-side helpers shared by capi.cpp and frame.cpp (HIP runtime API only, no kernels).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <string>

#include "../../include/zoic_amd.h"

namespace zoic {

// records the thread-local detail zoic_last_error_string() returns (defined in capi.cpp) and hands the status back
zoic_status fail_status(zoic_status s, const std::string &msg);

template <class T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// every entry point runs on the camera's device and leaves the calling thread's current device as it found it
class DeviceGuard {
    int prev_ = -1;
    bool switched_ = false;
    hipError_t err_ = hipSuccess;
public:
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != device) { err_ = hipSetDevice(device); switched_ = err_ == hipSuccess && prev_ >= 0; }
    }
    ~DeviceGuard() { if (switched_) (void)hipSetDevice(prev_); }
    hipError_t error() const { return err_; }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

}  // namespace zoic
