// obb_api.hip -- version / error plumbing of the C ABI (include/obbops.h).
#include <string.h>

#include "obb_common.h"

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

}  // namespace obb

extern "C" int obb_version(void) { return 100; }

extern "C" const char *obb_build_info(void) {
  return "libobbops 1.0 | OBB-GEOM v1 | gfx950 | -ffp-contract=off | HIP " __VERSION__;
}

extern "C" const char *obb_last_error(void) { return obb::err_buf(); }
