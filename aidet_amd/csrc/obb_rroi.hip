// obb_rroi.hip -- rotated RoI-Align forward / backward for gfx950.
//
// Rotated counterpart of ROIAlignForwardV1 / RoIAlignForwardV2
// (mmdet/ops/roi_align/src/roi_align_kernel.cu:64-117, roi_align_kernel_v2.cu:62-128)
// and RoIAlignBackwardFeatureV2 (roi_align_kernel_v2.cu:179-263).  Same bilinear
// tap rules (roi_align_kernel_v2.cu:15-60); the sampling grid is laid out in
// the RoI's own frame and rotated by theta about the RoI centre.
//
// Work split (the reference recomputes coordinates + weights for every output
// element, i.e. once per channel): here a thread owns one output bin and a
// group of kChanPerThread channels, computes each sample's 4 taps ONCE and
// reuses them across its channels; consecutive threads own consecutive bins of
// the same channel group, so output stores are contiguous runs of ph*pw floats
// and the feature gathers of a wave stay inside a few rows of a few planes
// (L1/L2 resident) -- NCHW is kept because the op must be a drop-in.
#include "obb_common.h"
#include "obb_geom.cuh"

namespace obb {

constexpr int kChanPerThread = 8;

struct RoiFrame {
  int b, gh, gw;
  float cx, cy, c, s, bin_w, bin_h, x0, y0, cnt;
};

__device__ __forceinline__ void make_frame(const float *__restrict__ roi, float scale, int ph,
                                           int pw, int sample_num, float min_size, RoiFrame &f) {
  f.b = (int)roi[0];
  f.cx = roi[1] * scale;
  f.cy = roi[2] * scale;
  float rw = roi[3] * scale, rh = roi[4] * scale;
  rw = max_sel(rw, min_size);
  rh = max_sel(rh, min_size);
  sincos_det(roi[5], f.s, f.c);
  f.bin_h = rh / (float)ph;
  f.bin_w = rw / (float)pw;
  f.gh = sample_num > 0 ? sample_num : (int)ceilf(rh / (float)ph);
  f.gw = sample_num > 0 ? sample_num : (int)ceilf(rw / (float)pw);
  f.x0 = -0.5f * rw;
  f.y0 = -0.5f * rh;
  int ci = f.gh * f.gw;
  f.cnt = (float)(ci < 1 ? 1 : ci);
}

__device__ __forceinline__ void sample_xy(const RoiFrame &f, int ph_i, int pw_i, int iy, int ix,
                                          float &x, float &y) {
  float ly = f.y0 + (float)ph_i * f.bin_h;
  float t1 = ((float)iy + 0.5f) * f.bin_h;
  ly = ly + t1 / (float)f.gh;
  float lx = f.x0 + (float)pw_i * f.bin_w;
  float t2 = ((float)ix + 0.5f) * f.bin_w;
  lx = lx + t2 / (float)f.gw;
  float xc = lx * f.c, ys = ly * f.s;
  float xs = lx * f.s, yc = ly * f.c;
  float rx = xc - ys, ry = xs + yc;
  x = f.cx + rx;
  y = f.cy + ry;
}

// tap offsets + weights; all-zero weights for a sample outside the map
__device__ __forceinline__ void bilinear_taps(int height, int width, float y, float x, int (&o)[4],
                                              float (&w)[4]) {
  const bool outside = (y < -1.0f) || (y > (float)height) || (x < -1.0f) || (x > (float)width);
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
  else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
  else x_high = x_low + 1;
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  o[0] = y_low * width + x_low;  o[1] = y_low * width + x_high;
  o[2] = y_high * width + x_low; o[3] = y_high * width + x_high;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
  if (outside) {
    o[0] = o[1] = o[2] = o[3] = 0;
    w[0] = w[1] = w[2] = w[3] = 0.0f;
  }
}

// grid: (K, channel chunks); block: 256 threads; item = (channel group, bin)
__global__ __launch_bounds__(256) void rroi_forward_kernel(
    const float *__restrict__ feat, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int groups_per_block,
    float *__restrict__ out) {
  const int64_t k = blockIdx.x;
  RoiFrame f;
  make_frame(rois + k * 6, scale, ph, pw, sample_num, min_size, f);
  const int nb = ph * pw;
  const int items = groups_per_block * nb;
  const int c_block = blockIdx.y * groups_per_block * kChanPerThread;
  const bool bad = f.b < 0 || f.b >= B;
  const int64_t plane = (int64_t)H * W;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int grp = it / nb, bin = it - grp * nb;
    const int c0 = c_block + grp * kChanPerThread;
    if (c0 >= C) continue;
    const int bi = bin / pw, bj = bin - bi * pw;
    float acc[kChanPerThread];
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c) acc[c] = 0.0f;
    if (!bad) {
      const float *base = feat + ((int64_t)f.b * C + c0) * plane;
      for (int iy = 0; iy < f.gh; ++iy) {
        for (int ix = 0; ix < f.gw; ++ix) {
          float x, y;
          sample_xy(f, bi, bj, iy, ix, x, y);
          int o[4];
          float w[4];
          bilinear_taps(H, W, y, x, o, w);
#pragma unroll
          for (int c = 0; c < kChanPerThread; ++c) {
            if (c0 + c < C) {
              const float *p = base + (int64_t)c * plane;
              float v = w[0] * p[o[0]];
              v = v + w[1] * p[o[1]];
              v = v + w[2] * p[o[2]];
              v = v + w[3] * p[o[3]];
              acc[c] = acc[c] + v;
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c)
      if (c0 + c < C) out[((k * C + c0 + c) * (int64_t)nb) + bin] = acc[c] / f.cnt;
  }
}

__global__ __launch_bounds__(256) void rroi_backward_kernel(
    const float *__restrict__ gout, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int groups_per_block,
    float *__restrict__ gfeat) {
  const int64_t k = blockIdx.x;
  RoiFrame f;
  make_frame(rois + k * 6, scale, ph, pw, sample_num, min_size, f);
  if (f.b < 0 || f.b >= B) return;
  const int nb = ph * pw;
  const int items = groups_per_block * nb;
  const int c_block = blockIdx.y * groups_per_block * kChanPerThread;
  const int64_t plane = (int64_t)H * W;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int grp = it / nb, bin = it - grp * nb;
    const int c0 = c_block + grp * kChanPerThread;
    if (c0 >= C) continue;
    const int bi = bin / pw, bj = bin - bi * pw;
    float g[kChanPerThread];
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c)
      g[c] = (c0 + c < C) ? gout[((k * C + c0 + c) * (int64_t)nb) + bin] : 0.0f;
    float *base = gfeat + ((int64_t)f.b * C + c0) * plane;
    for (int iy = 0; iy < f.gh; ++iy) {
      for (int ix = 0; ix < f.gw; ++ix) {
        float x, y;
        sample_xy(f, bi, bj, iy, ix, x, y);
        int o[4];
        float w[4];
        bilinear_taps(H, W, y, x, o, w);
        if (w[0] == 0.0f && w[1] == 0.0f && w[2] == 0.0f && w[3] == 0.0f) continue;
#pragma unroll
        for (int c = 0; c < kChanPerThread; ++c) {
          if (c0 + c < C) {
            float *p = base + (int64_t)c * plane;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              float d = g[c] * w[t];
              d = d / f.cnt;
              atomicAdd(p + o[t], d);
            }
          }
        }
      }
    }
  }
}

static int rroi_check(const char *who, int B, int C, int H, int W, int64_t K, int ph, int pw) {
  if (B < 0 || C < 0 || H < 0 || W < 0 || K < 0)
    return set_err(OBB_EINVAL, "%s: negative dimension", who);
  if (ph <= 0 || pw <= 0) return set_err(OBB_EINVAL, "%s: out size must be positive (%d x %d)", who, ph, pw);
  if ((int64_t)ph * pw > 4096) return set_err(OBB_EINVAL, "%s: out size %d x %d too large", who, ph, pw);
  if (K > 2147483647LL) return set_err(OBB_EINVAL, "%s: too many RoIs", who);
  return OBB_OK;
}

static void rroi_geometry(int C, int ph, int pw, int &groups_per_block, unsigned &chunks) {
  const int nb = ph * pw;
  int gpb = 256 / nb;
  if (gpb < 1) gpb = 1;
  if (gpb > 8) gpb = 8;
  const int groups = (C + kChanPerThread - 1) / kChanPerThread;
  if (gpb > groups) gpb = groups;
  groups_per_block = gpb;
  chunks = (unsigned)((groups + gpb - 1) / gpb);
}

}  // namespace obb

extern "C" int obb_rroi_align_forward_f32(const float *feat_dev, int B, int C, int H, int W,
                                          const float *rois_dev, int64_t K, int ph, int pw,
                                          float spatial_scale, int sample_num, float min_size,
                                          float *out_dev, void *stream) {
  using namespace obb;
  int rc = rroi_check("obb_rroi_align_forward_f32", B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (K == 0 || C == 0) return OBB_OK;
  if (!feat_dev || !rois_dev || !out_dev) return set_err(OBB_EINVAL, "obb_rroi_align_forward_f32: null pointer");
  if (H == 0 || W == 0 || B == 0) return set_err(OBB_EINVAL, "obb_rroi_align_forward_f32: empty feature map with %lld RoIs", (long long)K);
  int gpb; unsigned chunks;
  rroi_geometry(C, ph, pw, gpb, chunks);
  if (chunks > 65535u) return set_err(OBB_EINVAL, "obb_rroi_align_forward_f32: too many channels (%d)", C);
  hipLaunchKernelGGL(rroi_forward_kernel, dim3((unsigned)K, chunks), dim3(256), 0, (hipStream_t)stream,
                     feat_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, gpb, out_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_rroi_align_backward_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                           const float *rois_dev, int64_t K, int ph, int pw,
                                           float spatial_scale, int sample_num, float min_size,
                                           float *grad_feat_dev, void *stream) {
  using namespace obb;
  int rc = rroi_check("obb_rroi_align_backward_f32", B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (K == 0 || C == 0 || H == 0 || W == 0 || B == 0) return OBB_OK;
  if (!grad_out_dev || !rois_dev || !grad_feat_dev) return set_err(OBB_EINVAL, "obb_rroi_align_backward_f32: null pointer");
  int gpb; unsigned chunks;
  rroi_geometry(C, ph, pw, gpb, chunks);
  if (chunks > 65535u) return set_err(OBB_EINVAL, "obb_rroi_align_backward_f32: too many channels (%d)", C);
  hipLaunchKernelGGL(rroi_backward_kernel, dim3((unsigned)K, chunks), dim3(256), 0, (hipStream_t)stream,
                     grad_out_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, gpb, grad_feat_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
