// obb_api.hip -- version / error plumbing of the C ABI (include/obbops.h).
#include <string.h>

#include <atomic>

#include "obb_common.h"
#include "obb_nms_internal.h"

namespace obb {

static thread_local char g_err[512] = "";

char *err_buf() { return g_err; }

int set_err(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// (-1 = unset; the loader runs this before any caller can reach the library)
static std::atomic<int64_t> g_tune[OBB_TUNE_COUNT];
static const bool g_tune_init = [] { for (auto &t : g_tune) t.store(-1, std::memory_order_relaxed); return true; }();

int64_t tuning(int knob, int64_t def) {
  const int64_t v = g_tune[knob].load(std::memory_order_relaxed);
  return v < 0 ? def : v;
}

}  // namespace obb

extern "C" int obb_set_tuning(int knob, int64_t value) {
  if (knob < 0 || knob >= OBB_TUNE_COUNT) return obb::set_err(OBB_EINVAL, "obb_set_tuning: unknown knob %d", knob);
  obb::g_tune[knob].store(value < 0 ? -1 : value, std::memory_order_relaxed);
  if (knob == OBB_TUNE_NMS_GRID) obb::nms_grid_hints_clear();      // (the calling thread starts afresh: its "skip the grid" notes go)
  return OBB_OK;
}

extern "C" int obb_version(void) { return 104; }

extern "C" const char *obb_build_info(void) {
  return "libobbops 1.4 | OBB-GEOM v1 | gfx950 | -ffp-contract=off | HIP " __VERSION__;
}

extern "C" const char *obb_last_error(void) { return obb::err_buf(); }
