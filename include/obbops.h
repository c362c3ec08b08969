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
 *   kind OBB_KIND_HBB (obb_nms_*, obb_overlaps_f32): row = x1, y1, x2, y2 with the reference's +1 pixel
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
#define OBB_ERETRY (-4)     /* obb_nms_async_result / obb_tiled_result: the call's fast route gave up on this scene
                               (the uniform grid of the NMS: a crowded cell) -- repeat the call, or the tiled phase, as it
                               is: the library has noted it for the calling thread and takes the general route.  The
                               synchronous entry points repeat by themselves and never return this.                    */
#define OBB_ESTALLED (-5)   /* an NMS call's resolve kernel stopped at its bound without converging: its workgroups wait
                               for each other and must be resident together -- possible only when another persistent
                               kernel holds the device's slots.  The keep list is NOT valid; repeat the call.           */

#define OBB_KIND_RBOX 0
#define OBB_KIND_POLY 1
#define OBB_KIND_HBB 2  /* NMS and overlaps: row = x1,y1,x2,y2 (+score for NMS), the reference's horizontal boxes */
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
 *           rotated boxes -- and, with OBB_KIND_HBB, for the horizontal boxes themselves in the
 *           reference's own arithmetic (geometry.py:61-86: +1 widths, NaN / -0 results of
 *           degenerate boxes included).  a_dev (m, 5|8|4), b_dev (n, 5|8|4) row-major fp32.
 *   aligned == 0: out_dev is (m, n) row-major.
 *   aligned != 0: m must equal n, out_dev is (m).
 * m*n == 0 is a no-op returning OBB_OK (reference: geometry.py:54-55).
 * ------------------------------------------------------------------------- */
OBB_API int obb_overlaps_f32(const float *a_dev, int64_t m, const float *b_dev, int64_t n,
                     int kind, int mode, int aligned, float *out_dev, void *stream);
/* float64 horizontal boxes (m, 4) x (n, 4) -> double overlaps: bbox_overlaps is dtype-generic in the reference
 * (mmdet/core/bbox/geometry.py:47-88); same expressions as the OBB_KIND_HBB kind of obb_overlaps_f32, in double. */
OBB_API int obb_hbb_overlaps_f64(const double *a_dev, int64_t m, const double *b_dev, int64_t n, int mode, int aligned,
                                 double *out_dev, void *stream);

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
 *              call again with it (loop: which of the library's 64 internal pair lists a
 *              pair lands in depends on scheduling, so a retry can ask for more once
 *              again; every request at least doubles and 64 * n*(n-1)/2 always suffices).
 *              Capacity hints: the calling THREAD remembers, per kind and size class of n (its bit length), the
 *              capacity its last overflowing call needed; obb_nms_workspace_bytes(n, kind, 0) then sizes the default
 *              workspace for it, so a caller that always asks for the default repeats a dense scene's call once, not
 *              every time.  A fitting call that used less than 1/8 of a hinted capacity drops the hint.  (The only
 *              mutable state of the library besides the error string and the profile: thread-local, no locks.)
 *              Size: about 1.5 KB per box at the default capacity (64 pairs of 16 bytes, the 64-byte records, the
 *              sorted route's arrays) plus, for 256 <= n <= 2^24, the cell table of the uniform-grid route: 512 bytes
 *              per cell, about n / 2.5 cells (a power of two, at most 2^22 = 2 GiB) -- 150 MB at n = 1e5, 2 GB at
 *              1.25e6, 13.8 GB at 1e7.
 * Returns when the call's result record has arrived in host memory (the last act of its last kernel: keep_dev and K are
 * final; `stream` may still be retiring that kernel -- work queued behind it is ordered as usual).
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

/* Extended form: everything obb_nms_grouped_f32 does, plus `flags`:
 *   OBB_NMS_STRICT  suppress on overlap > iou_thr -- the comparison of the reference's CUDA
 *                   kernel (devIoU > thr, mmdet/ops/nms/src/nms_kernel.cu:61) -- instead of the
 *                   CPU reference's >= (nms_cpu.cpp:56), which is the default everywhere.  The
 *                   two only differ for pairs whose IoU EQUALS the threshold (integer boxes).
 * OBB_KIND_HBB rows are decided with the reference's own arithmetic (nms_cpu.cpp:19,47-56:
 * +1 widths, inter / (a_i + a_j - inter)), not with the polygon clip.  With that kind and
 * the >= rule, iou_thr <= 0 (or NaN) is served as nms_cpu.cpp:56 serves it: `ovr >= thr` then
 * holds for DISJOINT boxes too, every later box of a group falls to the first one unless its ovr
 * is NaN (two zero-area boxes) -- one all-pairs workgroup instead of the chain (round 5; refused
 * with OBB_EINVAL until round 4).  Per-box thresholds (thr_dev) must be positive under >=: the
 * library does not read them on the host.                                                    */
#define OBB_NMS_STRICT 1
/*   OBB_NMS_RETAIN  (round 6) the caller hands the SAME workspace to call after call and promises that nothing else writes
 *                   it in between (no other library call, no other thread, no other tensor in its memory).  The library
 *                   then keeps state in it: the tail of a call clears what the next call's first atomics count in and
 *                   leaves the frame of its own boxes, so the next call of the same size class (the table geometry: a
 *                   step function of n) does without the frame launch -- THREE launches instead of four on the grid
 *                   route below 2^18 boxes, ~7 us of a 56 us call at 100 000 boxes.  Results are the same bit for bit:
 *                   any frame gives the same keep list (coordinates clamp into it, pairs are only ever added); a scene
 *                   unlike its predecessor at worst finds a crowded grid and is repeated with a frame of its own.
 *                   The first call on a workspace, a call of another size class and a call after anything else wrote
 *                   the workspace SEED the state (the usual four launches).  The calling thread notes per workspace
 *                   pointer what its header should say, and every launch of a three-launch call checks the header
 *                   first: a workspace that another thread used or whose memory was handed out again in between makes
 *                   the call leave untouched and run again seeded -- but a caller that lets something else write the
 *                   COUNTERS and not the header breaks the promise, and the result with it.  Other routes (n <= 10 240,
 *                   n >= 2^18, a crowded scene) ignore the flag.  Synchronous entry point only (OBB_EINVAL elsewhere).
 *                   obb_nms_last_retained(): 0 the last call did nothing of the kind, 1 it seeded, 2 three launches. */
#define OBB_NMS_RETAIN 2
/* Ends what the calling thread notes about a workspace it used with OBB_NMS_RETAIN: call it before the workspace's memory is
 * freed or used for anything else (an allocator may hand the same address out again with the header intact and the counters
 * not).  NULL: all of the thread's notes.  Always OBB_OK. */
OBB_API int obb_nms_retain_forget(const void *workspace_dev);
OBB_API int obb_nms_ex_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                           const int32_t *group_dev, const float *thr_dev, int flags,
                           int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                           size_t workspace_bytes, void *stream);

/* float64 horizontal boxes: dets_dev (n, 5) DOUBLES [x1, y1, x2, y2, score].
 * Replaces: the double instantiation of nms_cpu_kernel<scalar_t> (AT_DISPATCH_FLOATING_TYPES,
 *           mmdet/ops/nms/src/nms_cpu.cpp:62-68; exercised by tests/test_nms.py:33-41 with float64 arrays).
 * Areas, intersection and `ovr >= threshold` are evaluated in double on the caller's doubles (the float threshold
 * promoted, as the reference's `const float threshold` is), "precedes" on the 64-bit key of the double score; only the
 * broad phase works on coordinates rounded to fp32 (conservatively).  Same keep / info / workspace contract as
 * obb_nms_ex_f32 with kind = OBB_KIND_HBB (workspace: obb_nms_workspace_bytes(n, OBB_KIND_HBB, capacity)). */
OBB_API int obb_nms_hbb_f64(const double *dets_dev, int64_t n, float iou_thr, const int32_t *group_dev,
                            const float *thr_dev, int flags, int64_t *keep_dev, int64_t *info_host,
                            void *workspace_dev, size_t workspace_bytes, void *stream);

/* Asynchronous form: enqueues the same kernels and returns WITHOUT synchronising `stream`
 * (the reference's nms_cuda copies the mask to the host and loops there, nms_kernel.cu:104-131;
 * a detector calls it once per FPN level and once per class, rpn_head.py:92, rbbox_nms.py:43 --
 * here those calls can be queued back to back and collected with one synchronisation).
 *   result_dev  OBB_NMS_RESULT_BYTES of device (or device-mapped host) memory; the last kernel
 *               writes the call's result record there.  int64 word 0 of the record is K, so a
 *               consumer on the same stream can use keep_dev[0 .. K) without the host.
 * After the caller has synchronised and copied the record to the host,
 * obb_nms_async_result(record, n, workspace_bytes, info_host) fills info_host[4] like
 * obb_nms_f32 and returns OBB_OK, or OBB_EWORKSPACE (info_host[3] = a capacity that suffices;
 * keep_dev is then invalid and the call must be repeated with a larger workspace).
 * Each in-flight call needs its own workspace and record unless they are on ONE stream.    */
#define OBB_NMS_RESULT_BYTES 64
OBB_API int obb_nms_async_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                              const int32_t *group_dev, const float *thr_dev, int flags,
                              int64_t *keep_dev, void *result_dev, void *workspace_dev,
                              size_t workspace_bytes, void *stream);
OBB_API int obb_nms_async_result(const void *result_host, int64_t n, size_t workspace_bytes,
                                 int64_t *info_host);

/* ---------------------------------------------------------------------------
 * Soft-NMS on horizontal boxes.
 * Replaces: at::Tensor soft_nms(dets, threshold, method, sigma, min_score)
 *           mmdet/ops/nms/src/nms_cpu.cpp:71-210 (CPU only in the reference,
 *           nms_wrapper.py:63-118), including its in-place array permutation, which decides
 *           ties between equal scores.
 *   dets_dev      (n, 5) x1, y1, x2, y2, score
 *   method        0 hard, 1 linear, 2 gaussian (the wrapper exposes 1 and 2, nms_wrapper.py:103)
 *   out_dets_dev  (n, 5): the K surviving boxes in selection order with their decayed scores
 *   out_inds_dev  int64[n]: their original indices
 *   count_host    K
 * One workgroup (the rounds are a dependency chain); synchronises `stream`.
 * method 2 calls the device expf: scores may differ from glibc's in the last ulp.
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_soft_nms_workspace_bytes(int64_t n);
OBB_API int obb_soft_nms_f32(const float *dets_dev, int64_t n, float iou_thr, int method, float sigma,
                             float min_score, float *out_dets_dev, int64_t *out_inds_dev,
                             int64_t *count_host, void *workspace_dev, size_t workspace_bytes,
                             void *stream);

/* float64 rows (n, 5): soft_nms_cpu_kernel<double> (AT_DISPATCH_FLOATING_TYPES, nms_cpu.cpp:212-215): overlaps, decayed scores
 * and the `< min_score` / `> iou_thr` comparisons in double (the float arguments promoted, :170,:184); out_dets_dev
 * (n, 5) doubles.  n <= 4096 (up to 2048 rows a kernel that needs no array positions, else -- or where two remaining boxes share
 * the maximum -- the state in registers with positions).  Synchronises `stream` (returns the count). */
OBB_API int obb_soft_nms_f64(const double *dets_dev, int64_t n, float iou_thr, int method, float sigma,
                             float min_score, double *out_dets_dev, int64_t *out_inds_dev,
                             int64_t *count_host, void *stream);

/* Soft-NMS of several INDEPENDENT groups without the host in between: workgroup g replays the reference's loop on the rows
 * [seg_dev[g], seg_dev[g + 1]) as an array of its own (two launches: a kernel that needs no array positions serves the groups
 * whose maxima are unique -- nearly all -- and the array kernel behind it the groups the first one marked; same rows either way).
 * Replaces: the per-class loop of multiclass_nms_with_index, mmdet/core/post_processing/rbbox_nms.py:29-49
 *           (one `soft_nms` call = one host round trip per class; a detection head has 15 classes).
 *   dets_dev      (n, 5), the rows of a group contiguous (class-major), seg_dev int64[num_groups + 1] ascending
 *   out_dets_dev / out_inds_dev  group g's survivors at rows seg[g] .. seg[g] + counts[g]; indices are relative
 *                 to the group's first row (what the per-class call returns)
 *   counts_dev    int64[num_groups] (DEVICE memory: the call does not synchronise)
 *   workspace     obb_soft_nms_workspace_bytes(n) bytes */
OBB_API int obb_soft_nms_grouped_f32(const float *dets_dev, int64_t n, const int64_t *seg_dev, int64_t num_groups,
                                     float iou_thr, int method, float sigma, float min_score,
                                     float *out_dets_dev, int64_t *out_inds_dev, int64_t *counts_dev,
                                     void *workspace_dev, size_t workspace_bytes, void *stream);
/* the same with the caller's bound on the rows of ONE group (the segment offsets are device memory): groups of up to 2048
 * rows keep their state in LDS whatever the total is.  A group larger than the bound gets count -1 (nothing written). */
OBB_API int obb_soft_nms_grouped_max_f32(const float *dets_dev, int64_t n, const int64_t *seg_dev, int64_t num_groups,
                                         int64_t max_group, float iou_thr, int method, float sigma, float min_score,
                                         float *out_dets_dev, int64_t *out_inds_dev, int64_t *counts_dev,
                                         void *workspace_dev, size_t workspace_bytes, void *stream);

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
 * RoI-Align over a feature PYRAMID in one launch.
 * Replaces: the per-level loop of SingleRoIExtractor.forward
 *           mmdet/models/roi_extractors/single_level.py:89-107 -- per level a boolean mask, a
 *           gather of the RoIs, a kernel launch and a scatter into the (K, C, h, w) result.
 *   feat_devs_host  [num_levels] device pointers to (B, C, H_l, W_l) fp32 maps (HOST array)
 *   H_host, W_host, scale_host  [num_levels] map sizes and spatial scales (1 / stride)
 *   roi_level_dev   int32[K] level of every RoI (single_level.py:54-73 map_roi_levels; computed
 *                   by the caller so that the level borders are decided by the SAME float
 *                   operations as the reference's); may be NULL when num_levels == 1.
 *                   A level outside [0, num_levels) yields zeros for that RoI.
 *   roi_kind        0: rois (K, 6) rotated, 1: (K, 5) horizontal V1, 2: (K, 5) V2 aligned
 *   out_dev         (K, C, ph, pw), every RoI written in place: no per-level copies
 * min_size applies to roi_kind 0 only (obb_rroi_align_forward_f32).
 * ------------------------------------------------------------------------- */
#define OBB_MAX_LEVELS 8
OBB_API int obb_roi_align_levels_forward_f32(const float *const *feat_devs_host, const int *H_host,
                                             const int *W_host, const float *scale_host, int num_levels,
                                             int B, int C, const float *rois_dev,
                                             const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                             int sample_num, int roi_kind, float min_size,
                                             float *out_dev, void *stream);
/* SingleRoIExtractor.map_roi_levels, mmdet/models/roi_extractors/single_level.py:54-73, as one launch:
 * levels_dev[k] = clamp(floor(log2(sqrt(area_k) / finest_scale + 1e-6)), 0, num_levels - 1), area = w * h of a rotated
 * RoI (roi_kind 0: (K, 6) [b, cx, cy, w, h, theta]) or (x2 - x1 + 1)(y2 - y1 + 1) (kinds 1, 2: (K, 5)). */
OBB_API int obb_roi_map_levels_i32(const float *rois_dev, int64_t K, int roi_kind, float finest_scale, int num_levels,
                                   int32_t *levels_dev, void *stream);

/* ---------------------------------------------------------------------------
 * RoI-Align forward (pyramid or single map, any roi_kind) with a caller-provided WORKSPACE of
 * obb_roi_forward_workspace_bytes(K) bytes: the arguments and results of
 * obb_roi_align_levels_forward_f32, plus one small launch that puts the RoIs into the Morton order of
 * their centres -- the blocks then visit them in that order, RoIs that run at the same time share
 * footprint lines in an XCD's L2 and the kernel fetches each map byte about once (configs[2]: 189 -> 70 MB
 * from the fabric).  Which block computes a RoI never changes a value; out_dev is in the caller's RoI order.
 * workspace_dev == NULL is allowed (no ordering = obb_roi_align_levels_forward_f32).
 * Replaces: the same reference code as obb_roi_align_levels_forward_f32 (RoIAlign V1 / V2 kernels,
 *           roi_align_kernel.cu:64-141, roi_align_kernel_v2.cu:62-128; SingleRoIExtractor.forward
 *           single_level.py:89-107) -- none of which takes a workspace (THCudaMalloc scratch is the
 *           reference's convention for ops that need one, nms_kernel.cu:94).
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_roi_forward_workspace_bytes(int64_t K);
OBB_API int obb_roi_align_forward_ws_f32(const float *const *feat_devs_host, const int *H_host,
                                         const int *W_host, const float *scale_host, int num_levels,
                                         int B, int C, const float *rois_dev,
                                         const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                         int sample_num, int roi_kind, float min_size,
                                         float *out_dev, void *workspace_dev, size_t workspace_bytes,
                                         void *stream);

/* ---------------------------------------------------------------------------
 * RoI-Align forward on CHANNELS-LAST maps: feat_devs_host[l] points at a (B, H_l, W_l, C) fp32 array -- the memory of
 * a PyTorch (B, C, H, W) tensor in torch.channels_last -- everything else (RoI kinds, levels, sampling and tap rules,
 * the (K, C, ph, pw) output in the reference's layout) is obb_roi_align_forward_ws_f32; the workspace arguments are
 * accepted for that signature and not used (ordering the RoIs does not pay on this layout).  Values are bit-equal to that entry point on the
 * permuted map.  C % 4 == 0, 16-byte aligned maps, H * W * C < 2^30 per image; OBB_EINVAL otherwise (the Python wrapper
 * then copies the map to NCHW, as the reference's `features.contiguous()` would).
 * Replaces: RoIAlign V1 / V2 forward (roi_align_kernel.cu:64-141, roi_align_kernel_v2.cu:62-128), which read NCHW
 *           only (roi_align_cuda.cpp:37-42: CHECK_CONTIGUOUS) -- a channels-last trunk pays a 4 B H W C-byte copy there.
 * obb_rroi_align_forward_nhwc_f32: one map, rotated (K, 6) RoIs = obb_rroi_align_forward_f32 on a channels-last map.
 * ------------------------------------------------------------------------- */
OBB_API int obb_roi_align_forward_nhwc_f32(const float *const *feat_devs_host, const int *H_host,
                                           const int *W_host, const float *scale_host, int num_levels,
                                           int B, int C, const float *rois_dev,
                                           const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                           int sample_num, int roi_kind, float min_size,
                                           float *out_dev, void *workspace_dev, size_t workspace_bytes,
                                           void *stream);
OBB_API int obb_rroi_align_forward_nhwc_f32(const float *feat_dev, int B, int C, int H, int W,
                                            const float *rois_dev, int64_t K, int ph, int pw,
                                            float spatial_scale, int sample_num, float min_size,
                                            float *out_dev, void *stream);

/* ---------------------------------------------------------------------------
 * RoI-Align backward as a GATHER (any roi_kind: 0 rotated (K,6) rows, 1 RoIAlign V1, 2 V2-aligned
 * (K,5) rows), with a caller-provided workspace of obb_roi_backward_workspace_bytes(K) bytes.
 * Replaces: the same reference kernels as obb_rroi_align_backward_f32 / obb_roi_align_backward_f32
 *           (ROIAlignBackwardV1 roi_align_kernel.cu:188-280, RoIAlignBackwardFeatureV2
 *           roi_align_kernel_v2.cu:179-263, both atomicAdd scatters).
 * Differences from those two entry points: grad_feat is OVERWRITTEN (every element of the
 * (B,C,H,W) map is stored exactly once: no zero fill by the caller), no atomics are used for RoIs
 * of up to 2047 footprint pixels / 256 samples / 64 bins, and the result is reproducible bit for bit
 * (fixed summation order: RoI index, then tap number).  Larger RoIs are added by the atomics kernel.
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_roi_backward_workspace_bytes(int64_t K);
OBB_API int obb_roi_backward_gather_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                        const float *rois_dev, int64_t K, int roi_kind, int ph, int pw,
                                        float spatial_scale, int sample_num, float min_size,
                                        float *grad_feat_dev, void *workspace_dev, size_t workspace_bytes,
                                        void *stream);

/* ---------------------------------------------------------------------------
 * theta-OBB -> 4 corner points in cv2.boxPoints order.
 * Replaces: thetaobb2pointobb, mmdet/core/rbbox/transforms.py:45-55 (per-box
 *           cv2 call on the host).  in (n,5) -> out (n,8).
 * ------------------------------------------------------------------------- */
OBB_API int obb_rbox_to_poly_f32(const float *rbox_dev, int64_t n, float *poly_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Row plumbing of the tiled-inference merge (BASELINE.json configs[4]): what sits between the
 * per-patch NMS and the merged per-(image, class) NMS.
 * Replaces: the text round trip of mmdet/datasets/dota.py:197-336 -- patch offsets parsed from
 *           the file name and added to the coordinates (:262-283, 302-305), per-class merge
 *           thresholds (:320-331) -- and the host-side list surgery around it
 *           (mmdet/apis/test.py:116-146).  All arrays are device pointers; int64 index arrays
 *           as torch produces them.  Values are computed exactly like the tensor expressions
 *           of aidet_amd/dota_merge.py (add, then divide by `rate`, in fp32).
 *
 * obb_patch_groups_i32:        groups[i] = (int32)(patch_index[i] * num_classes + labels[i])
 * obb_patch_rows_to_image_f32: rows[i] = { (cx + ox) / rate, (cy + oy) / rate, w / rate, h / rate,
 *                              theta, score, label, image } of detection keep[i] (keep = NULL:
 *                              i) with (ox, oy) = patch_xy[patch_index], image =
 *                              patch_image[patch_index]; dets (n,6), rows (k,8).  An index out
 *                              of range gives a zero row with label = image = -1.
 * obb_merge_rows_split_f32:    rows (n, cols = 8 | 11 | 7: box + score, label, image) ->
 *                              boxes (n, cols - 2), groups[i] = (int32)(image * num_classes +
 *                              label), thrs[i] = class_thrs[label] (boxes / thrs / class_thrs may
 *                              be NULL; a label outside [0, num_classes) takes default_thr).
 * obb_gather_rows_f32:         out[i, :out_cols] = src[idx[i], :out_cols]   (src (n, cols), idx (k,) int64,
 *                              out (k, out_cols), 1 <= out_cols <= cols)
 * ------------------------------------------------------------------------- */
OBB_API int obb_patch_groups_i32(const int64_t *labels_dev, const int64_t *patch_index_dev, int64_t n,
                                 int64_t num_classes, int32_t *groups_dev, void *stream);
OBB_API int obb_patch_rows_to_image_f32(const float *dets_dev, const int64_t *labels_dev,
                                        const int64_t *patch_index_dev, int64_t n,
                                        const int64_t *keep_dev, int64_t k, const float *patch_xy_dev,
                                        const int64_t *patch_image_dev, int64_t num_patches, float rate,
                                        float *rows_dev, void *stream);
OBB_API int obb_merge_rows_split_f32(const float *rows_dev, int64_t n, int cols, int64_t num_classes,
                                     const float *class_thrs_dev, float default_thr, float *boxes_dev,
                                     int32_t *groups_dev, float *thrs_dev, void *stream);
/* the same with one more output / a row pitch: patch_of_row_dev int64[k] (may be NULL) = the patch index of every
 * output row (-1 for a source row out of range) -- the order key of the multi-rank merge; rows of `ld` >= cols floats
 * (rows that carry more columns than the merge reads, e.g. that key) */
OBB_API int obb_patch_rows_to_image_keyed_f32(const float *dets_dev, const int64_t *labels_dev,
                                              const int64_t *patch_index_dev, int64_t n,
                                              const int64_t *keep_dev, int64_t k, const float *patch_xy_dev,
                                              const int64_t *patch_image_dev, int64_t num_patches, float rate,
                                              float *rows_dev, int64_t *patch_of_row_dev, void *stream);
OBB_API int obb_merge_rows_split_ld_f32(const float *rows_dev, int64_t n, int cols, int ld, int64_t num_classes,
                                        const float *class_thrs_dev, float default_thr, float *boxes_dev,
                                        int32_t *groups_dev, float *thrs_dev, void *stream);
OBB_API int obb_gather_rows_f32(const float *src_dev, int64_t n, int cols, const int64_t *idx_dev, int64_t k,
                                int out_cols, float *out_dev, void *stream);

/* ---------------------------------------------------------------------------
 * Row routing and order restoration of the multi-rank merge (config 5).
 * Replaces: the pickle / all_gather / interleave of collect_results, mmdet/apis/test.py:116-146 (every rank's
 *           results to every rank, `ordered_results.extend(list(res))` restoring the dataset order), as far as the
 *           rows stay on the device: which rows travel where and in which order they are merged.
 *
 * obb_rows_route_f32:   rows (n, cols) -> out (n, cols + 1): the rows in stable buckets by destination
 *     ((int64)rows[i, id_col] mod world, the sign of the divisor), bucket 0 first, each with one more column holding
 *     (float)order_key[i] (0 without keys); counts_dev int64[world + 1]: rows per bucket, and in [world] the number
 *     of rows whose key is SMALLER than their predecessor's (the order restoration below needs 0).
 *     world <= 64.  Three launches: per-block bucket counts, one scan, the scatter.
 * obb_rows_restore_f32: the rows of `world` sources, source r in rows [seg_start_host[r], seg_start_host[r + 1])
 *     (the last one ends at n), each source in non-decreasing key (column key_col, integers as float) -> out
 *     (total, out_cols): all rows with 0 <= key <= key_max in ascending key; equal keys: ascending source, then the
 *     source's order (a stable merge).  Rows with a key outside [0, key_max] are padding and dropped; `total` = the
 *     number of real rows (the caller knows it from the exchanged counts).  (key_max + 1) * world <=
 *     OBB_RESTORE_MAX_SLOTS.  A source that is not in non-decreasing key order: *status_dev (device-visible int32,
 *     may be NULL; written only then) = the number of offending rows, the output is unspecified.
 *     One memset + three launches.
 * ------------------------------------------------------------------------- */
#define OBB_RESTORE_MAX_SLOTS (1 << 20)
OBB_API size_t obb_rows_route_workspace_bytes(int64_t n, int world);
OBB_API int obb_rows_route_f32(const float *rows_dev, int64_t n, int cols, int id_col, int world,
                               const int64_t *order_key_dev, float *out_dev, int64_t *counts_dev,
                               void *workspace_dev, size_t workspace_bytes, void *stream);
OBB_API size_t obb_rows_restore_workspace_bytes(int world, int64_t key_max);
OBB_API int obb_rows_restore_f32(const float *rows_dev, int64_t n, int cols, int key_col, const int64_t *seg_start_host,
                                 int world, int64_t key_max, int out_cols, int64_t total, float *out_dev,
                                 int32_t *status_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------
 * One rank-side PHASE of a tiled pass (config 5) per call, without the host in between (round 4).
 * Replaces: what a rank does between the detector and the collectives of multi_gpu_test / collect_results_gpu
 *           (mmdet/apis/test.py:31-146) and what rank 0 does after them in DOTADataset.merge_txt
 *           (mmdet/datasets/dota.py:310-336: per image and class, thresholds :321-331) -- as TWO calls that each
 *           enqueue all their kernels and return without synchronising.  The kept count of a phase's NMS is read by
 *           the following kernels from the NMS result record on the device.
 *
 * obb_tiled_local_f32: the n detections of this rank's patches (dets (n, 6) theta-OBB + score in patch pixels,
 *   labels / patch_index int64[n]) -> per-patch, per-class rotated NMS (group = patch * num_classes + label, derived
 *   inside the NMS's first kernel) -> the K kept rows in image coordinates, rows_dev (n, 9) capacity:
 *   [(cx + ox) / rate, (cy + oy) / rate, w / rate, h / rate, theta, score, label, image, patch] -- in keep order
 *   (world == 1) or in stable buckets by destination rank `image mod world` (2 <= world <= 64), bucket 0 first.
 *   counts_dev: int64[world + 2] + OBB_NMS_RESULT_BYTES: rows per destination | rows whose patch index is smaller
 *   than their predecessor's (0 for a rank that visits its patches in order) | K | the NMS's result record.
 * obb_tiled_merge_f32: rows received from `world` sources (recv_dev (m, ld), source r = rows [seg_start_host[r],
 *   seg_start_host[r + 1]); columns [box + score | label | image | ... | key]) -> (key_max >= 0) the `total_rows`
 *   real rows in ascending key = column ld - 1, a stable merge of the sources (obb_rows_restore_f32) into ordered_dev
 *   (total_rows, ld) -> per-(image, class) NMS of those rows read in place (group = image * num_classes + label and
 *   thr = class_thrs_dev[label] derived inside the NMS's first kernel; cols 8 / 11 / 7 = theta-OBB / polygon /
 *   horizontal rows) -> keep_dev int64 (ascending, into the ordered rows) and kept_dev (K, out_cols) = their first
 *   out_cols columns.  key_max < 0: the rows are used as they are (total_rows == m).  result_dev: the NMS's
 *   result record (device), K in word 0.
 * obb_tiled_result(record copied to the host, phase 0 local / 1 merge, the sizes the phase was called with):
 *   info_host[4] like obb_nms_f32; OBB_EWORKSPACE when the phase's pair lists overflowed (repeat the phase with a
 *   workspace for pair_capacity >= info_host[3]).
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_tiled_local_workspace_bytes(int64_t n, int world, int64_t pair_capacity);
OBB_API int obb_tiled_local_f32(const float *dets_dev, const int64_t *labels_dev, const int64_t *patch_index_dev,
                                int64_t n, const float *patch_xy_dev, const int64_t *patch_image_dev,
                                int64_t num_patches, int64_t num_classes, float rate, float iou_thr, int world,
                                float *rows_dev, int64_t *counts_dev, void *workspace_dev, size_t workspace_bytes,
                                void *stream);
OBB_API size_t obb_tiled_merge_workspace_bytes(int64_t total_rows, int cols, int world, int64_t key_max,
                                               int64_t pair_capacity);
OBB_API int obb_tiled_merge_f32(const float *recv_dev, int64_t m, int cols, int ld, const int64_t *seg_start_host,
                                int world, int64_t key_max, int64_t total_rows, int64_t num_classes,
                                const float *class_thrs_dev, float iou_thr, int flags, float *ordered_dev,
                                float *kept_dev, int out_cols, int64_t *keep_dev, void *result_dev,
                                int32_t *status_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
OBB_API int obb_tiled_result(const void *record_host, int phase, int64_t n, int cols, int world, int64_t key_max,
                             size_t workspace_bytes, int64_t *info_host);

/* ---------------------------------------------------------------------------
 * Proposal / detection SELECTION stages of the detector (config 4), between a head's convolutions and its NMS.
 * Replaces: the per-level tensor-op chain of RPNHead.get_bboxes_single, mmdet/models/anchor_heads/rpn_head.py:55-104
 *           (sigmoid :71, per-level topk :77-81, delta2bbox :82-83 = mmdet/core/bbox/transforms.py:34-111, the
 *           `proposals[:nms_post]` cut :96 and the final topk :100-103) -- ~40 small launches around one NMS call.
 *
 * obb_rpn_select_decode_f32: per level (one workgroup each) the `nms_pre` best anchors by score (all of a level with
 *   at most nms_pre anchors, in anchor order), decoded with delta2bbox and clamped to (max_h, max_w) (max_h <= 0: no
 *   clamp) -> dets_dev (N, 5) [x1, y1, x2, y2, score], levels_dev int32[N], level-major, descending score inside a
 *   selected level (equal scores: ascending anchor index; torch.topk leaves that order unspecified).
 *   N = sum_l min(nms_pre, H_l W_l A) is returned in *num_rows_host (no synchronisation: it depends on shapes only;
 *   call once with dets_dev = NULL to size the outputs).  cls / reg / anchors: HOST arrays of device pointers to the
 *   (A * cls_out, H, W) logits, the (4 A, H, W) deltas and the (H W A, 4) anchors of every level; use_sigmoid 1:
 *   cls_out = 1, score = sigmoid(logit); 0: cls_out = 2, score = softmax(.)[1] (rpn_head.py:68-75).
 *   nms_pre <= 2048.
 * obb_rows_topk_f32: rows[keep] (keep ascending int64, as an NMS returns it) -> optional per-level cut (the rows
 *   are level-major, level l = rows [level_base[l], level_base[l+1]); at most nms_post kept rows per level, in keep
 *   order) -> the `top` (<= 2048) best by column score_col, descending, equal scores by ascending row -> out_dev
 *   (count, cols), out_idx_dev int64[count] (source rows; may be NULL), count_dev int64[1] (device).
 * ------------------------------------------------------------------------- */
OBB_API int obb_rpn_select_decode_f32(const float *const *cls_devs_host, const float *const *reg_devs_host,
                                      const float *const *anchor_devs_host, const int *H_host, const int *W_host,
                                      int num_levels, int num_anchors, int use_sigmoid, int nms_pre,
                                      const float *means_host, const float *stds_host, float wh_ratio_clip,
                                      int max_h, int max_w, float *dets_dev, int32_t *levels_dev,
                                      int64_t *num_rows_host, void *stream);
/* the same with a workspace (obb_rpn_select_workspace_bytes; 0 = none needed): levels of >= 32 768 anchors are walked
 * by one workgroup per 8192 anchors (obb::rpn_collect_kernel) before the level's workgroup selects among the
 * candidates they leave -- one more launch, 0.32 -> 0.1 ms for a 5-level 1024^2 pyramid.  Same results, except which
 * members of a TIE at the nms_pre-th score are taken (anchor order here; torch.topk leaves it open). */
OBB_API size_t obb_rpn_select_workspace_bytes(const int *H_host, const int *W_host, int num_levels, int num_anchors,
                                              int nms_pre);
OBB_API int obb_rpn_select_decode_ws_f32(const float *const *cls_devs_host, const float *const *reg_devs_host,
                                         const float *const *anchor_devs_host, const int *H_host, const int *W_host,
                                         int num_levels, int num_anchors, int use_sigmoid, int nms_pre,
                                         const float *means_host, const float *stds_host, float wh_ratio_clip,
                                         int max_h, int max_w, float *dets_dev, int32_t *levels_dev,
                                         int64_t *num_rows_host, void *workspace_dev, size_t workspace_bytes,
                                         void *stream);
OBB_API int obb_rows_topk_f32(const float *rows_dev, int64_t num_rows, int cols, int score_col,
                              const int64_t *keep_dev, int64_t K, const int *level_base_host, int num_levels,
                              int nms_post, int top, float *out_dev, int64_t *out_idx_dev, int64_t *count_dev,
                              void *stream);

/* Detection heads: what sits between a head's FC outputs and its class-wise NMS.
 * Replaces: BBoxHead.get_det_bboxes up to the NMS, mmdet/models/bbox_heads/bbox_head.py:139-166 (softmax :149,
 *           delta2bbox of EVERY class of every RoI :152-153, rescale :160-166), the candidate lists of
 *           multiclass_nms_with_index (mmdet/core/post_processing/rbbox_nms.py:22-43: per class
 *           `scores[:, i] > score_thr`), RBBoxHead.get_det_rbboxes_parallel (rbbox_head.py:253-296: delta2thetaobb,
 *           mmdet/core/rbbox/transforms.py:356-395) and thetaobb_nms_by_bbox_nms (rbbox_nms.py:64-119).
 *
 * obb_head_select_decode_f32: rois (n, 5) [b, x1, y1, x2, y2], cls_score (n, num_classes) logits (class 0 =
 *   background), pred (n, pred_cols) deltas with pred_cols = dim * num_classes (class specific) or dim (class
 *   agnostic); dim 4 = delta2bbox (+ clamp to (max_h, max_w) when max_h > 0), dim 5 = delta2thetaobb; scale_factor
 *   > 0 divides the box (not the angle) by it (`rescale=True`).  Output: the (RoI, class) pairs with softmax
 *   probability > score_thr, CLASS-MAJOR, RoIs ascending inside a class (the order of nonzero() on the (class, RoI)
 *   mask): dets_dev (M, dim + 1) decoded box + score, rows_dev / labels_dev int32[M] (RoI index, 0-based foreground
 *   class), counts_dev int64[num_classes]: candidates per foreground class, then M (device memory: no
 *   synchronisation).  Capacity of the outputs: n * (num_classes - 1) rows.  n <= 2048, num_classes <= 17.
 * obb_head_gather_decode_f32: the same decode + softmax score of ANOTHER head for the (rows[p], labels[p]) pairs
 *   p = sel[j] (sel = NULL: p = j), j < K -> dets_dev (K, dim + 1). */
OBB_API int obb_head_select_decode_f32(const float *rois_dev, const float *cls_score_dev, const float *pred_dev,
                                       int64_t n, int num_classes, int pred_cols, int dim, float score_thr,
                                       const float *means_host, const float *stds_host, float wh_ratio_clip,
                                       int max_h, int max_w, float scale_factor, float *dets_dev,
                                       int32_t *rows_dev, int32_t *labels_dev, int64_t *counts_dev, void *stream);
OBB_API int obb_head_gather_decode_f32(const float *rois_dev, const float *cls_score_dev, const float *pred_dev,
                                       int num_classes, int pred_cols, int dim, const int32_t *rows_dev,
                                       const int32_t *labels_dev, const int64_t *sel_dev, int64_t K,
                                       const float *means_host, const float *stds_host, float wh_ratio_clip,
                                       int max_h, int max_w, float scale_factor, float *dets_dev, void *stream);

/* ---------------------------------------------------------------------------
 * RoI-Align backward as a gather on CHANNELS-LAST maps: grad_out_dev is (K, C, ph, pw) in the reference's layout,
 * grad_feat_dev a (B, H, W, C) fp32 array (the memory of a torch.channels_last (B, C, H, W) tensor), OVERWRITTEN like
 * obb_roi_backward_gather_f32's: every element stored exactly once, no atomics for planned RoIs, summation order (RoI
 * index, tap number) -- bit-equal to obb_roi_backward_gather_f32 on the permuted map, reproducible run to run.
 * Workspace: obb_roi_backward_nhwc_workspace_bytes(K, C, ph, pw) (the plan + the transposed gradient, 4 K C ph pw bytes).
 * C % 4 == 0, K <= 65535, 16-byte aligned pointers.
 * Replaces: ROIAlignBackwardV1 (roi_align_kernel.cu:188-280) / RoIAlignBackwardFeatureV2 (roi_align_kernel_v2.cu:179-263),
 *           which scatter with atomicAdd into an NCHW map only.
 * ------------------------------------------------------------------------- */
OBB_API size_t obb_roi_backward_nhwc_workspace_bytes(int64_t K, int C, int ph, int pw);
OBB_API int obb_roi_backward_gather_nhwc_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                             const float *rois_dev, int64_t K, int roi_kind, int ph, int pw,
                                             float spatial_scale, int sample_num, float min_size,
                                             float *grad_feat_dev, void *workspace_dev, size_t workspace_bytes,
                                             void *stream);

/* ---------------------------------------------------------------------------
 * Test / tuning hooks.  NOT part of the drop-in surface (no reference counterpart): the
 * parity tests use them to force every size through every code path, and the experiment
 * scripts under tools/ to sweep launch geometry.  Process-wide; value < 0 restores the
 * built-in default.  The library reads no environment variable.
 * ------------------------------------------------------------------------- */
#define OBB_TUNE_NMS_FUSED_MAX 0      /* largest n served by the single-workgroup kernel (<= 1024)   */
#define OBB_TUNE_NMS_DIRECT_MAX 1     /* largest n served without the spatial sort (<= 65536)        */
#define OBB_TUNE_NMS_SCAN_BLOCKS 2    /* grid of the sorted route's scan kernel                      */
#define OBB_TUNE_RROI_GROUPS 3        /* 8-channel groups per RoI-Align block                        */
#define OBB_TUNE_OVERLAPS_ROWS 4      /* rows per block of the pairwise-overlap kernel (8..64)       */
#define OBB_TUNE_NMS_INLINE_MAX 5     /* largest n whose sorted scan clips its candidates in line     */
#define OBB_TUNE_RROI_ORDER_MIN 6     /* fewest RoIs the workspace forward puts into Morton order      */
#define OBB_TUNE_RROI_GATHER_CHAN 7   /* channels per block of the gather backward (16 or 32)          */
#define OBB_TUNE_RPN_BARRIER_TICKS 8  /* bound (100 MHz ticks) of the level barrier of the RPN selection; 0 forces the fallback */
#define OBB_TUNE_SOFT_NMS_REG 9       /* float Soft-NMS: 1 the position-tracking register kernel (the float64 one; one array, n <= 8192); 2 the array kernel only (no position-free first try; one array and grouped calls) */
#define OBB_TUNE_NMS_FUSED_COMPACT 10  /* 0: the sorted route's compaction as a launch of its own (as from 2^18 boxes on); 2: the waiting shards give up at once */
#define OBB_TUNE_RROI_PLAN 11          /* fewest RoIs the workspace forward of RoI-Align plans per RoI (default 1024; 0 = always) */
#define OBB_TUNE_NMS_GRID 12           /* grid route of the NMS: 0 never, 1 wherever it applies, 2 the same but it gives up at once (test hook for the repeat paths); default auto.  Setting it clears the calling thread's "skip the grid" notes */
#define OBB_TUNE_SOFT_NMS_SHAPE 13     /* 1: the position-free Soft-NMS kernel as 256 threads x up to 8 boxes (default: up to 1024 threads x 1-4 boxes; <= 64 boxes: one wave); 3: never the one-wave kernel; 4: the one-wave kernel up to 512 boxes */
#define OBB_TUNE_RROI_NHWC_NQ 14       /* channels-last RoI-Align forward: lanes per tap = slab of 4 x NQ channels per block (8, 16, 32, 64) */
#define OBB_TUNE_RROI_NHWC_XCD 15      /* (unused since the channels-last forward stopped ordering its RoIs; kept so that the knob numbers stay put) */
#define OBB_TUNE_RROI_NHWC_WAVES 16   /* channels-last RoI-Align forward: waves per block (4 or 8) */
#define OBB_TUNE_NMS_WATCH_RECORD 17   /* 0: the synchronous NMS entry point sleeps in hipStreamSynchronize instead of watching its pinned result record */
#define OBB_TUNE_RROI_NHWC_GATHER_WAVES 18   /* channels-last gather backward: waves per map tile (4, 8, 16) */
#define OBB_TUNE_NMS_RESOLVE_TAIL 19   /* resolve, bit 0: the last <= 4 edges per thread of a shard are retired from registers (1, default) or swept through memory (0); bit 1: the first sweep uses agent-scope loads like the others (measurement only); bit 2: the resolve reports itself stalled (test hook for OBB_ESTALLED) */
#define OBB_TUNE_NMS_RESOLVE_PARTS 20   /* resolve of calls of >= 2^18 boxes: workgroups per edge shard (1 .. 16; never more than half of what the device keeps resident) */
#define OBB_TUNE_NMS_RETAIN 21          /* 0: OBB_NMS_RETAIN is ignored (every call runs the frame launch): A/B measurements */
#define OBB_TUNE_COUNT 22
OBB_API int obb_set_tuning(int knob, int64_t value);
/* which route the calling thread's last synchronous obb_nms_*_f32 call took: 0 fused (one workgroup), 1 direct, 2 sorted,
 * 4 grid, 5 grid gave up (crowded cell) and the call was repeated on the sorted route; -1 none yet */
OBB_API int obb_nms_last_route(void);
/* what the calling thread's last obb_nms_ex_f32 call did with its workspace's retained state (OBB_NMS_RETAIN): 0 nothing,
 * 1 seeded it (four launches), 2 ran on it (three launches) */
OBB_API int obb_nms_last_retained(void);
/* which kernel served the calling thread's last obb_soft_nms_f32 call: 2 the position-free kernel (n <= 4096, unique
 * maxima, no NaN), 3 that kernel gave up and the array kernel replayed the reference's permutation, 0 the array kernel
 * alone, 1 the register kernel (knob); -1 none yet */
OBB_API int obb_soft_nms_last_path(void);

#ifdef __cplusplus
}
#endif
#endif /* OBBOPS_H_ */
