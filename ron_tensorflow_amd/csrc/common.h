// Shared host-side helpers of libron_hip.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
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

}  // namespace ron
