/* A plain C99 host of the C ABI: the header must be valid C (no C++ / torch types) and the
 * library must link and validate arguments without a GPU.  Built and run by
 * tests/test_abi_and_host.py::test_header_is_c99_and_library_links_from_c. */
#include "obbops.h"

#include <stdio.h>
#include <string.h>

int main(void) {
  int64_t info[4] = {7, 7, 7, 7};
  if (obb_version() < 100) return 1;
  if (!strstr(obb_build_info(), "gfx950")) return 2;
  if (obb_nms_f32(0, -1, OBB_KIND_RBOX, 0.5f, 0, info, 0, 0, 0) != OBB_EINVAL) return 3;
  if (!strstr(obb_last_error(), "negative")) return 4;
  if (info[0] != 0) return 5;                                   /* outputs are cleared first */
  if (obb_nms_f32(0, 0, OBB_KIND_HBB, 0.5f, 0, info, 0, 0, 0) != OBB_OK) return 6;   /* n = 0: nothing to do */
  if (obb_nms_workspace_bytes(100000, OBB_KIND_RBOX, 0) == 0) return 7;
  if (obb_overlaps_f32(0, 0, 0, 5, OBB_KIND_RBOX, OBB_MODE_IOU, 0, 0, 0) != OBB_OK) return 8;   /* empty */
  if (obb_overlaps_f32(0, 3, 0, 5, 9, OBB_MODE_IOU, 0, 0, 0) != OBB_EINVAL) return 9;            /* bad kind */
  /* the row plumbing of the tiled merge: argument checks run without a GPU */
  if (obb_patch_groups_i32(0, 0, 0, 15, 0, 0) != OBB_OK) return 10;                                  /* n = 0 */
  if (obb_patch_rows_to_image_f32(0, 0, 0, 5, 0, 5, 0, 0, 1, 0.0f, 0, 0) != OBB_EINVAL) return 11;   /* rate <= 0 */
  if (obb_merge_rows_split_f32(0, 4, 9, 15, 0, 0.3f, 0, 0, 0, 0) != OBB_EINVAL) return 12;           /* 9 columns */
  if (obb_gather_rows_f32(0, 4, 8, 0, 0, 8, 0, 0) != OBB_OK) return 13;                              /* k = 0 */
  if (obb_gather_rows_f32(0, 4, 8, 0, 2, 9, 0, 0) != OBB_EINVAL) return 14;                          /* out_cols > cols */
  if (obb_roi_backward_workspace_bytes(512) == 0) return 15;
  if (obb_roi_backward_gather_f32(0, 1, 1, 8, 8, 0, 4, 7, 7, 7, 0.25f, 2, 1.0f, 0, 0, 0, 0) != OBB_EINVAL) return 16;   /* roi_kind 7 */
  printf("c abi ok: %s\n", obb_build_info());
  return 0;
}
