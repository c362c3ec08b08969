// obb_common.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/obbops.h"

namespace obb {

// thread-local message behind obb_last_error()
char *err_buf();
int set_err(int code, const char *fmt, ...);

#define OBB_HIP_TRY(expr)                                                          \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return obb::set_err(OBB_EHIP, "%s failed: %s (%s:%d)", #expr,                \
                          hipGetErrorString(e_), __FILE__, __LINE__);              \
  } while (0)

#define OBB_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess)                                                          \
      return obb::set_err(OBB_EHIP, "kernel launch failed: %s (%s:%d)",            \
                          hipGetErrorString(e_), __FILE__, __LINE__);              \
  } while (0)

// obb_set_tuning() values (obb_api.hip); `def` when the knob is unset
int64_t tuning(int knob, int64_t def);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace obb
