/*
 * obbops.h -- C ABI of libobbops.so: oriented-bounding-box operators for
 * MI355X (gfx950), hand-written HIP.
 *
 * Drop-in boundary for the OBB hot path of AIDet / MMDetection v1.x.  Each entry
 * point names the reference interface it replaces (paths relative to the
 * reference checkout, jwwangchn/aidet).  The reference binds its native ops
 * through pybind11 modules imported next to the Python wrappers
 * (mmdet/ops/nms/nms_wrapper.py:4 `from . import nms_cpu, nms_cuda`); the
 * replacement binds this library with ctypes (aidet_amd/_lib.py) -- see
 * INTEGRATION.md for the stub a maintainer adds.
 *
 * Conventions
 *   - every pointer named *_dev is DEVICE memory of the current HIP device;
 *     *_host is host memory.  No torch types, no C++ types.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Kernels are enqueued on it; only obb_nms_f32 synchronises it (to return
 *     the keep count, like the reference's nms_cuda which is synchronous by
 *     construction: mmdet/ops/nms/src/nms_kernel.cu:104-131).
 *   - the library never allocates device memory: the caller passes outputs and
 *     a workspace sized by the *_workspace_bytes query (reference: scratch from
 *     the caching allocator, nms_kernel.cu:94,133).
 *     (Host side: obb_nms_* keeps ONE 256-byte pinned host buffer per calling
 *     thread, into which the last kernel writes the result counters.)
 *   - return value: OBB_OK or a negative OBB_E* code; obb_last_error() gives a
 *     thread-local message (reference: AT_ASSERTM/AT_CHECK -> RuntimeError,
 *     mmdet/ops/nms/src/nms_cpu.cpp:7, roi_align_cuda.cpp:37-42).
 *   - all arithmetic is fp32 ("OBB-GEOM v1"); inputs are never
 *     modified.
 *
 * Box encodings
 *   kind OBB_KIND_RBOX: row = cx, cy, w, h, theta   (theta-OBB, radians, the
 *        cv2.boxPoints meaning used by mmdet/core/rbbox/transforms.py:45-55)
 *   kind OBB_KIND_POLY: row = x1,y1,x2,y2,x3,y3,x4,y4 (convex quad, any
 *        orientation; the pointobb of transforms.py:45-55)
 *   kind OBB_KIND_HBB (obb_nms_* only): row = x1, y1, x2, y2 with the reference's +1 pixel
 *        convention (nms_cpu.cpp:19,52-55): the box is the theta = 0 OBB with centre
 *        ((x1+x2)/2, (y1+y2)/2), w = x2-x1+1, h = y2-y1+1 -- the continuous box
 *        [x1-.5, x2+.5] x [y1-.5, y2+.5], whose intersection widths are the reference's.
 *   NMS rows carry one extra trailing float: the score.
 */
#ifndef OBBOPS_H_
#define OBBOPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define OBB_API __attribute__((visibility("default")))
#else
#define OBB_API
#endif

#define OBB_OK 0
#define OBB_EINVAL (-1)     /* bad argument                                   */
#define OBB_EWORKSPACE (-2) /* workspace too small; see obb_nms_f32           */
#define OBB_EHIP (-3)       /* a HIP runtime call failed                      */

#define OBB_KIND_RBOX 0
#define OBB_KIND_POLY 1
#define OBB_KIND_HBB 2  /* NMS only: row = x1,y1,x2,y2 (+score), the reference's horizontal boxes */
#define OBB_MODE_IOU 0
#define OBB_MODE_IOF 1

/* library / ABI version (major*100 + minor) and build description */
OBB_API int obb_version(void);
OBB_API const char *obb_build_info(void);
OBB_API const char *obb_last_error(void);

/* ---------------------------------------------------------------------------
 * Pairwise overlaps.
 * Replaces: bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)
 *           mmdet/core/bbox/geometry.py:4-88  (pure-torch HBB IoU/IoF), for
 *           rotated boxes.  a_dev (m, 5|8), b_dev (n, 5|8) row-major fp32.
 *   aligned == 0: out_dev is (m, n) row-major.
 *   aligned != 0: m must equal n, out_dev is (m).
 * m*n == 0 is a no-op returning OBB_OK (reference: geometry.py:54-55).
 * ------------------------------------------------------------------------- */
OBB_API int obb_overlaps_f32(const float *a_dev, int64_t m, const float *b_dev, int64_t n,
                     int kind, int mode, int aligned, float *out_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Greedy rotated / polygon NMS.
 * Replaces: at::Tensor nms(const at::Tensor& dets, const float threshold)
 *           mmdet/ops/nms/src/nms_cuda.cpp:8-17 -> nms_kernel.cu:71-139 and
 *           mmdet/ops/nms/src/nms_cpu.cpp:62-68, for rotated boxes.
 * Semantics (those of the CPU reference, nms_cpu.cpp:21-59): visit boxes by
 * descending score (ties: lower index first); a visited, unsuppressed box i
 * suppresses every later j with overlap(i,j) >= iou_thr; the result is the
 * ASCENDING list of original indices of unsuppressed boxes.
 *
 *   dets_dev   (n, 6) for OBB_KIND_RBOX, (n, 9) for OBB_KIND_POLY, (n, 5) for OBB_KIND_HBB
 *   keep_dev   int64[n]; the first K entries are valid on return
 *   info_host  int64[4] (may be NULL): [0] K, [1] overlap edges found,
 *              [2] resolve rounds, [3] candidate pairs clipped
 *   workspace  >= obb_nms_workspace_bytes(n, kind, pair_capacity) bytes.
 *              pair_capacity = number of candidate pairs (boxes whose AABBs
 *              meet) the workspace can hold; pass 0 for the default
 *              (max(256K, 64 n)).  If more exist the call returns
 *              OBB_EWORKSPACE with info_host[3] = a capacity that suffices;
 *              call again with it.  n*(n-1)/2 * 1.25 always suffices.
 * Synchronises `stream` before returning.
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_nms_workspace_bytes(int64_t n, int kind, int64_t pair_capacity);
OBB_API int obb_nms_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                size_t workspace_bytes, void *stream);

/* Grouped variant: boxes with different group_dev[i] never suppress each other
 * (per-class / per-image NMS in ONE call).  Replaces the coordinate-offset trick
 * of multiclass_nms (mmdet/core/post_processing/bbox_nms.py:56-63), the Python
 * loop over classes of rbbox_nms.py:29-49 and the per-class-file merge of
 * mmdet/datasets/dota.py:310-336.  group_dev int32[n] or NULL; thr_dev float[n]
 * or NULL gives one threshold per box -- the threshold of the EARLIER box of a
 * pair applies (per-class thresholds, dota.py:321-331; boxes of one group are
 * expected to share it).  Otherwise identical to obb_nms_f32.                  */
OBB_API int obb_nms_grouped_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                                const int32_t *group_dev, const float *thr_dev,
                                int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                                size_t workspace_bytes, void *stream);

/* Optional per-phase device timing of obb_nms_f32 (HIP events recorded on the
 * caller's stream; thread-local).  Phases: 0 clear + prep + quantise, 1 spatial
 * counting sort, 2 permute + tile culling, 3 scan (broad phase), 4 clip (narrow
 * phase), 5 resolve, 6 compaction.  No reference counterpart (the reference has
 * no kernel-level profiling).                                    */
#define OBB_NMS_PHASES 7
OBB_API int obb_nms_profile_enable(int level); /* 0 off, 1 scan kernel only, 2 all phases */
OBB_API int obb_nms_profile_last(float *phase_ms_host /* [OBB_NMS_PHASES] */);

/* ---------------------------------------------------------------------------
 * Rotated RoI-Align.
 * Replaces (for rotated RoIs): roi_align_cuda.forward_v1 / forward_v2 and
 *           backward_v1 / backward_v2, mmdet/ops/roi_align/src/roi_align_cuda.cpp:44-137
 *           (kernels roi_align_kernel.cu:64-117, roi_align_kernel_v2.cu:62-128,
 *           backward :179-263).
 *   feat_dev  (B, C, H, W) fp32 NCHW
 *   rois_dev  (K, 6) = batch_index, cx, cy, w, h, theta (image coordinates)
 *   out_dev   (K, C, ph, pw)
 *   spatial_scale multiplies cx, cy, w, h;  sizes are clamped to >= min_size
 *   (1 = RoIAlignForwardV2 aligned=False, roi_align_kernel_v2.cu:87-90;
 *    0 = V1's fmaxf(.,0), roi_align_kernel.cu:85-86);
 *   sample_num > 0: fixed sampling grid, else ceil(bin size) per axis.
 * theta = 0 reproduces RoIAlignForwardV2(aligned=False) on the equivalent
 * xyxy RoI.  A batch index outside [0, B) yields zeros for that RoI.
 * backward accumulates (atomic adds) into grad_feat_dev, which the caller
 * zero-fills first (reference: roi_align.py:63-64 new_zeros).
 * ------------------------------------------------------------------------- */
OBB_API int obb_rroi_align_forward_f32(const float *feat_dev, int B, int C, int H, int W,
                               const float *rois_dev, int64_t K, int ph, int pw,
                               float spatial_scale, int sample_num, float min_size,
                               float *out_dev, void *stream);
OBB_API int obb_rroi_align_backward_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                const float *rois_dev, int64_t K, int ph, int pw,
                                float spatial_scale, int sample_num, float min_size,
                                float *grad_feat_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Horizontal RoI-Align on the same kernels (theta = 0; the RoI row is decoded in the kernel).
 * Replaces: roi_align_cuda.forward / backward behind RoIAlignFunction,
 *           mmdet/ops/roi_align/roi_align.py:9-73 (needs THC on CUDA: not buildable on ROCm)
 *   rois_dev  (K, 5) batch index, x1, y1, x2, y2 (image scale, roi_align.py:136)
 *   aligned   0: the legacy V1 kernel's geometry (roi_align_kernel.cu:79-86: +1 widths, the
 *                module's default, roi_align.py:28-36)
 *             1: V2 with aligned=True (roi_align_kernel_v2.cu:79-86: half-pixel shift, no
 *                lower clamp of the size)
 * Everything else as obb_rroi_align_*.
 * ------------------------------------------------------------------------- */
OBB_API int obb_roi_align_forward_f32(const float *feat_dev, int B, int C, int H, int W,
                              const float *rois_dev, int64_t K, int ph, int pw,
                              float spatial_scale, int sample_num, int aligned,
                              float *out_dev, void *stream);
OBB_API int obb_roi_align_backward_f32(const float *grad_out_dev, int B, int C, int H, int W,
                               const float *rois_dev, int64_t K, int ph, int pw,
                               float spatial_scale, int sample_num, int aligned,
                               float *grad_feat_dev, void *stream);

/* ---------------------------------------------------------------------------
 * theta-OBB -> 4 corner points in cv2.boxPoints order.
 * Replaces: thetaobb2pointobb, mmdet/core/rbbox/transforms.py:45-55 (per-box
 *           cv2 call on the host).  in (n,5) -> out (n,8).
 * ------------------------------------------------------------------------- */
OBB_API int obb_rbox_to_poly_f32(const float *rbox_dev, int64_t n, float *poly_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OBBOPS_H_ */
