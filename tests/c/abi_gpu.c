/* A plain C99 host that RUNS the hot path on the GPU through the C ABI alone -- no Python, no torch, no C++: what any
 * FFI (cgo, JNI, N-API) would do.  Device memory and the stream come from the HIP runtime's C API; the results are
 * compared with values the CPU oracle produced (tests/c/abi_gpu_vectors.h, written by make_abi_gpu_vectors.py):
 *   obb_nms_f32       300 rotated boxes  -> keep indices, bit-exact
 *   obb_overlaps_f32  24 x 40 rotated boxes -> IoU, bit-identical floats
 *   obb_rroi_align_forward_nhwc_f32   6 RoIs on a channels-last 8 x 20 x 24 map -> within 1e-4
 * Built and run by tests/test_round5_gpu.py::test_c_host_runs_the_gpu_through_the_abi. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "obbops.h"
#include "abi_gpu_vectors.h"

#define HIP_TRY(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 100; } } while (0)

static float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(void) {
  hipStream_t st;
  HIP_TRY(hipSetDevice(0));
  HIP_TRY(hipStreamCreate(&st));
  /* ---- rotated NMS */
  {
    float *dets; int64_t *keep; void *ws; int64_t info[4]; int64_t host[NMS_N];
    const size_t wsb = obb_nms_workspace_bytes(NMS_N, OBB_KIND_RBOX, 0);
    int i;
    HIP_TRY(hipMalloc((void **)&dets, sizeof(nms_dets)));
    HIP_TRY(hipMalloc((void **)&keep, sizeof(int64_t) * NMS_N));
    HIP_TRY(hipMalloc(&ws, wsb));
    HIP_TRY(hipMemcpy(dets, nms_dets, sizeof(nms_dets), hipMemcpyHostToDevice));
    if (obb_nms_f32(dets, NMS_N, OBB_KIND_RBOX, 0.5f, keep, info, ws, wsb, st) != OBB_OK) { printf("nms: %s\n", obb_last_error()); return 1; }
    if (info[0] != NMS_K) { printf("nms kept %lld, oracle %d\n", (long long)info[0], NMS_K); return 2; }
    HIP_TRY(hipMemcpy(host, keep, sizeof(int64_t) * NMS_K, hipMemcpyDeviceToHost));
    for (i = 0; i < NMS_K; ++i) if (host[i] != nms_keep[i]) { printf("nms keep[%d] = %lld, oracle %lld\n", i, (long long)host[i], (long long)nms_keep[i]); return 3; }
    /* the same call with OBB_NMS_RETAIN (obbops.h): 300 boxes run on a route that keeps no state in the workspace -- the flag is
     * accepted, the result is the same, obb_nms_last_retained() says 0 */
    if (obb_nms_ex_f32(dets, NMS_N, OBB_KIND_RBOX, 0.5f, NULL, NULL, OBB_NMS_RETAIN, keep, info, ws, wsb, st) != OBB_OK ||
        info[0] != NMS_K || obb_nms_last_retained() != 0) { printf("nms with OBB_NMS_RETAIN: %s\n", obb_last_error()); return 8; }
    HIP_TRY(hipFree(dets)); HIP_TRY(hipFree(keep)); HIP_TRY(hipFree(ws));
  }
  /* ---- rotated IoU */
  {
    float *a, *b, *out; uint32_t host[IOU_M * IOU_N]; int i;
    HIP_TRY(hipMalloc((void **)&a, sizeof(iou_a))); HIP_TRY(hipMalloc((void **)&b, sizeof(iou_b)));
    HIP_TRY(hipMalloc((void **)&out, sizeof(host)));
    HIP_TRY(hipMemcpy(a, iou_a, sizeof(iou_a), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b, iou_b, sizeof(iou_b), hipMemcpyHostToDevice));
    if (obb_overlaps_f32(a, IOU_M, b, IOU_N, OBB_KIND_RBOX, OBB_MODE_IOU, 0, out, st) != OBB_OK) { printf("overlaps: %s\n", obb_last_error()); return 4; }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
    for (i = 0; i < IOU_M * IOU_N; ++i) if (host[i] != iou_ref[i]) { printf("iou[%d] = %g, oracle %g\n", i, as_float(host[i]), as_float(iou_ref[i])); return 5; }
    HIP_TRY(hipFree(a)); HIP_TRY(hipFree(b)); HIP_TRY(hipFree(out));
  }
  /* ---- rotated RoI-Align on a channels-last map */
  {
    float *feat, *rois, *out; static uint32_t host[ROI_K * ROI_C * 49]; int i; double worst = 0.0;
    HIP_TRY(hipMalloc((void **)&feat, sizeof(roi_feat_nhwc))); HIP_TRY(hipMalloc((void **)&rois, sizeof(roi_rois)));
    HIP_TRY(hipMalloc((void **)&out, sizeof(host)));
    HIP_TRY(hipMemcpy(feat, roi_feat_nhwc, sizeof(roi_feat_nhwc), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(rois, roi_rois, sizeof(roi_rois), hipMemcpyHostToDevice));
    if (obb_rroi_align_forward_nhwc_f32(feat, 1, ROI_C, ROI_H, ROI_W, rois, ROI_K, 7, 7, 0.25f, 2, 1.0f, out, st) != OBB_OK) { printf("rroi: %s\n", obb_last_error()); return 6; }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
    for (i = 0; i < ROI_K * ROI_C * 49; ++i) { const double d = fabs((double)as_float(host[i]) - (double)as_float(roi_ref[i])); if (d > worst) worst = d; }
    if (!(worst <= 1e-4)) { printf("rroi: max |diff| %g\n", worst); return 7; }
    HIP_TRY(hipFree(feat)); HIP_TRY(hipFree(rois)); HIP_TRY(hipFree(out));
  }
  HIP_TRY(hipStreamDestroy(st));
  printf("c host on the gpu ok: %s\n", obb_build_info());
  return 0;
}
