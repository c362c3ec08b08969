// obb_merge.hip -- row plumbing of the tiled-inference merge (BASELINE.json configs[4]).
//
// Between the two grouped NMS calls of a tiled pass (per-patch NMS -> merged per-image NMS) the
// reference round-trips through Python lists, text files and the external wwtool merge
// (mmdet/datasets/dota.py:197-336: patch offsets from the file name, translate, per-class
// thresholds).  As tensor ops that is ~20 small elementwise / index launches per pass over
// 1.25 M rows (0.4 ms of a 1.5 ms pass on one MI355X, a third of it one index_select); here
// three streaming kernels, each touching every row once:
//   obb_patch_groups_i32         group id of the per-patch NMS      (patch * classes + label)
//   obb_patch_rows_to_image_f32  kept rows -> image coordinates + label + image id
//   obb_merge_rows_split_f32     merge rows -> boxes | group ids | per-class thresholds
//   obb_gather_rows_f32          rows[idx]
// Pure data movement plus one add and one divide per coordinate, evaluated exactly like the
// tensor expressions they replace (aidet_amd/dota_merge.py): results are bit-identical.
#include "obb_common.h"

namespace obb {

constexpr int kMergeThreads = 256;

__global__ __launch_bounds__(kMergeThreads) void patch_groups_kernel(const int64_t *__restrict__ labels,
                                                                     const int64_t *__restrict__ patch_index,
                                                                     int64_t n, int64_t num_classes,
                                                                     int32_t *__restrict__ groups) {
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i < n) groups[i] = (int32_t)(patch_index[i] * num_classes + labels[i]);   // (wraps like Tensor.to(int32))
}

// out[i] = { (cx + ox) / rate, (cy + oy) / rate, w / rate, h / rate, theta, score, label, image }
// of detection keep[i] (or i); a row whose source or patch index is out of range becomes zeros
// with label = image = -1 (it then forms a group of its own and is easy to spot).
__global__ __launch_bounds__(kMergeThreads) void patch_rows_kernel(const float *__restrict__ dets,
                                                                   const int64_t *__restrict__ labels,
                                                                   const int64_t *__restrict__ patch_index,
                                                                   const int64_t *__restrict__ keep, int64_t k,
                                                                   int64_t n, const float *__restrict__ patch_xy,
                                                                   const int64_t *__restrict__ patch_image,
                                                                   int64_t num_patches, float rate,
                                                                   float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= k) return;
  const int64_t src = keep ? keep[i] : i;
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = make_float4(0.0f, 0.0f, -1.0f, -1.0f);
  if (src >= 0 && src < n) {
    const int64_t p = patch_index[src];
    if (p >= 0 && p < num_patches) {
      const float2 *d = reinterpret_cast<const float2 *>(dets + src * 6);   // rows of 24 bytes: 8-byte aligned
      const float2 d0 = d[0], d1 = d[1], d2 = d[2];
      const float2 o = reinterpret_cast<const float2 *>(patch_xy)[p];
      a.x = (d0.x + o.x) / rate; a.y = (d0.y + o.y) / rate;
      a.z = d1.x / rate; a.w = d1.y / rate;
      b.x = d2.x; b.y = d2.y;
      b.z = (float)labels[src];
      b.w = (float)patch_image[p];
    }
  }
  float4 *o4 = reinterpret_cast<float4 *>(out + i * 8);
  o4[0] = a; o4[1] = b;
}

// rows (n, nbox + 2): [box + score | label | image]  ->  boxes (n, nbox), group = image * classes + label,
// thr = class_thrs[label] (labels outside [0, classes) take `default_thr`)
__global__ __launch_bounds__(kMergeThreads) void merge_split_kernel(const float *__restrict__ rows, int cols,
                                                                    int64_t n, int64_t num_classes,
                                                                    const float *__restrict__ class_thrs,
                                                                    float default_thr,
                                                                    float *__restrict__ boxes,
                                                                    int32_t *__restrict__ groups,
                                                                    float *__restrict__ thrs) {
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= n) return;
  const int nbox = cols - 2;
  const float *r = rows + i * cols;
  if (boxes) {
    float *b = boxes + i * nbox;
    for (int c = 0; c < nbox; ++c) b[c] = r[c];
  }
  const int64_t label = (int64_t)r[nbox], image = (int64_t)r[nbox + 1];   // (truncation, like Tensor.long())
  groups[i] = (int32_t)(image * num_classes + label);
  if (thrs) thrs[i] = (class_thrs && label >= 0 && label < num_classes) ? class_thrs[label] : default_thr;
}

// out[i, :out_cols] = src[idx[i], :out_cols]; one thread per 4-byte word, consecutive threads = consecutive words
__global__ __launch_bounds__(kMergeThreads) void gather_rows_kernel(const float *__restrict__ src, int cols,
                                                                    int64_t n, const int64_t *__restrict__ idx,
                                                                    int64_t k, int out_cols, float *__restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (w >= k * out_cols) return;
  const int64_t i = w / out_cols;
  const int c = (int)(w - i * out_cols);
  const int64_t s = idx[i];
  out[w] = (s >= 0 && s < n) ? src[s * cols + c] : 0.0f;
}

static int merge_grid(const char *who, int64_t items, unsigned &blocks) {
  const int64_t b = (items + kMergeThreads - 1) / kMergeThreads;
  if (b > 2147483647LL) return set_err(OBB_EINVAL, "%s: too many rows", who);
  blocks = (unsigned)b;
  return OBB_OK;
}

}  // namespace obb

extern "C" int obb_patch_groups_i32(const int64_t *labels_dev, const int64_t *patch_index_dev, int64_t n,
                                    int64_t num_classes, int32_t *groups_dev, void *stream) {
  if (n < 0) return obb::set_err(OBB_EINVAL, "obb_patch_groups_i32: negative n");
  if (n == 0) return OBB_OK;
  if (!labels_dev || !patch_index_dev || !groups_dev) return obb::set_err(OBB_EINVAL, "obb_patch_groups_i32: null pointer");
  unsigned blocks;
  int rc = obb::merge_grid("obb_patch_groups_i32", n, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(obb::patch_groups_kernel, dim3(blocks), dim3(obb::kMergeThreads), 0, (hipStream_t)stream,
                     labels_dev, patch_index_dev, n, num_classes, groups_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_patch_rows_to_image_f32(const float *dets_dev, const int64_t *labels_dev,
                                           const int64_t *patch_index_dev, int64_t n,
                                           const int64_t *keep_dev, int64_t k, const float *patch_xy_dev,
                                           const int64_t *patch_image_dev, int64_t num_patches, float rate,
                                           float *rows_dev, void *stream) {
  if (n < 0 || k < 0 || num_patches < 0) return obb::set_err(OBB_EINVAL, "obb_patch_rows_to_image_f32: negative size");
  if (!(rate > 0.0f)) return obb::set_err(OBB_EINVAL, "obb_patch_rows_to_image_f32: rate must be positive");
  if (k == 0) return OBB_OK;
  if (!dets_dev || !labels_dev || !patch_index_dev || !patch_xy_dev || !patch_image_dev || !rows_dev)
    return obb::set_err(OBB_EINVAL, "obb_patch_rows_to_image_f32: null pointer");
  unsigned blocks;
  int rc = obb::merge_grid("obb_patch_rows_to_image_f32", k, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(obb::patch_rows_kernel, dim3(blocks), dim3(obb::kMergeThreads), 0, (hipStream_t)stream,
                     dets_dev, labels_dev, patch_index_dev, keep_dev, k, n, patch_xy_dev, patch_image_dev,
                     num_patches, rate, rows_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_merge_rows_split_f32(const float *rows_dev, int64_t n, int cols, int64_t num_classes,
                                        const float *class_thrs_dev, float default_thr, float *boxes_dev,
                                        int32_t *groups_dev, float *thrs_dev, void *stream) {
  if (n < 0) return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: negative n");
  if (cols != 7 && cols != 8 && cols != 11)
    return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: rows have 8 (theta-OBB), 11 (polygon) or 7 (horizontal) columns, got %d", cols);
  if (n == 0) return OBB_OK;
  if (!rows_dev || !groups_dev) return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: null pointer");
  unsigned blocks;
  int rc = obb::merge_grid("obb_merge_rows_split_f32", n, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(obb::merge_split_kernel, dim3(blocks), dim3(obb::kMergeThreads), 0, (hipStream_t)stream,
                     rows_dev, cols, n, num_classes, class_thrs_dev, default_thr, boxes_dev, groups_dev, thrs_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_gather_rows_f32(const float *src_dev, int64_t n, int cols, const int64_t *idx_dev, int64_t k,
                                   int out_cols, float *out_dev, void *stream) {
  if (n < 0 || k < 0 || cols <= 0 || out_cols <= 0 || out_cols > cols) return obb::set_err(OBB_EINVAL, "obb_gather_rows_f32: bad size");
  if (k == 0) return OBB_OK;
  if (!src_dev || !idx_dev || !out_dev) return obb::set_err(OBB_EINVAL, "obb_gather_rows_f32: null pointer");
  unsigned blocks;
  int rc = obb::merge_grid("obb_gather_rows_f32", k * out_cols, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(obb::gather_rows_kernel, dim3(blocks), dim3(obb::kMergeThreads), 0, (hipStream_t)stream,
                     src_dev, cols, n, idx_dev, k, out_cols, out_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
