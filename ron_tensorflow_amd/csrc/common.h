// Shared host-side helpers of libron_hip.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ron_hip.h"

namespace ron {

// Thread-local message behind ron_last_error().
void set_error(const char* fmt, ...);

#define RON_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ron::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                     __LINE__);                                                          \
      return RON_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define RON_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ron::set_error(__VA_ARGS__);        \
      return RON_ERR_INVALID;             \
    }                                     \
  } while (0)

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Dry run (RON_PLAN_ONLY=1 in the environment, read once): every host-side decision of the library is made - launch plans, tile
// choices, split-K factors, kernel arguments, workspace sizes - but no HIP call: device "allocations" are distinct fake addresses
// nobody dereferences, copies, events and kernel launches are skipped.  What `make asan` (csrc/Makefile) runs the host planners
// under AddressSanitizer / UBSan with, on a machine without a GPU (tools/plan_sweep.cpp, tests/test_asan_plan_sweep.py).
bool plan_only();
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
hipError_t dev_memset(void* p, int v, size_t bytes);
hipError_t dev_memset_async(void* p, int v, size_t bytes, hipStream_t s);
hipError_t dev_memcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t dev_set_device(int device);
hipError_t launch_error();                     // hipGetLastError(), hipSuccess in a dry run
template <class T> hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }
#define RON_LAUNCH(...)                                          \
  do {                                                           \
    if (!ron::plan_only()) hipLaunchKernelGGL(__VA_ARGS__);      \
  } while (0)

// hipFuncSetAttribute applies to the current device only: one of these per kernel instantiation remembers on which
// devices (id < 64) the dynamic-LDS limit has been raised.  Safe from several host threads: nobody returns before the
// attribute is set on his device.
struct PerDeviceOnce {
  std::atomic<uint64_t> done{0};
  std::mutex mu;
  hipError_t max_dynamic_lds(const void* kernel, int bytes) {
    if (plan_only()) return hipSuccess;
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev <= 63;     // unknown device: just set it again
    const uint64_t bit = known ? 1ull << dev : 0;
    if (known && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    std::lock_guard<std::mutex> lock(mu);
    if (known && (done.load(std::memory_order_relaxed) & bit)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && known) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// ron_post_cfg::input_flags, internal bit (not in include/ron_hip.h; reserved there): the workspace's counters are zero on entry and
// ron_post_np leaves them zero (topk_nms_kernel cleans up after itself) - set by ron_detect for its context-owned workspace only
constexpr unsigned kPostWsClean = 0x40000000u;

// class ids the post-processing kernels group and encode (a power of two: label masks, anchor * kMaxClasses + label keys)
constexpr int kMaxClasses = RON_MAX_CLASSES;
static_assert((kMaxClasses & (kMaxClasses - 1)) == 0, "kMaxClasses must be a power of two");

// Makes `device` current for the lifetime of the guard and restores the caller's device afterwards.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    if (plan_only()) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

}  // namespace ron
