// obb_rroi_internal.h -- device helpers shared by the RoI-Align translation units (obb_rroi.hip: NCHW maps,
// obb_rroi_nhwc.hip: channels-last maps): the RoI frame of the three RoI kinds, the sampling expressions and the
// bilinear tap rules of roi_align_kernel_v2.cu:15-60 / roi_align_kernel.cu:17-60.  Not part of the C ABI.
#pragma once
#include "obb_common.h"
#include "obb_geom.h"

namespace obb {

struct RoiFrame {
  int b, gh, gw, kind;
  float cnt;                      // divisor of a bin's sum
  float cx, cy, c, s, x0, y0;     // kind 0: centre, rotation, start in the RoI's own frame
  float xs, ys;                   // kinds 1, 2: start on the map
  float rw, rh, bin_w, bin_h;
};

// RoI row layouts.  kind 0: [b, cx, cy, w, h, theta].  Horizontal RoIs [b, x1, y1, x2, y2]
// (roi_align.py:136) are evaluated with the reference's own expressions:
//   kind 1 (legacy V1, roi_align_kernel.cu:79-86): start = x1*s, end = (x2+1)*s, size =
//          fmaxf(end - start, 0); the sum is divided by gh*gw (0/0 = NaN for an empty grid)
//   kind 2 (V2 aligned=True, roi_align_kernel_v2.cu:79-92): start = x1*s - 0.5, size = end -
//          start unclamped; divided by max(gh*gw, 1)
__device__ __forceinline__ int roi_stride(int kind) { return kind == 0 ? 6 : 5; }

__device__ __forceinline__ int grid_of(float bin, int sample_num) {
  if (sample_num > 0) return sample_num;
  const float g = ceilf(bin);
  return g > 65536.0f ? 65536 : (g > 0.0f ? (int)g : 0);   // (an empty or negative grid samples nothing)
}

__device__ __forceinline__ void make_frame(const float *__restrict__ roi, int kind, float scale, int ph,
                                           int pw, int sample_num, float min_size, RoiFrame &f) {
  f.b = (int)roi[0];
  f.kind = kind;
  f.cx = f.cy = f.x0 = f.y0 = f.xs = f.ys = 0.0f;
  f.c = 1.0f; f.s = 0.0f;
  if (kind == 0) {
    f.cx = roi[1] * scale;
    f.cy = roi[2] * scale;
    f.rw = max_sel(roi[3] * scale, min_size);
    f.rh = max_sel(roi[4] * scale, min_size);
    sincos_det(roi[5], f.s, f.c);
    f.x0 = -0.5f * f.rw;
    f.y0 = -0.5f * f.rh;
  } else if (kind == 1) {
    f.xs = roi[1] * scale;
    f.ys = roi[2] * scale;
    const float xe = (roi[3] + 1.0f) * scale, ye = (roi[4] + 1.0f) * scale;
    f.rw = fmaxf(xe - f.xs, 0.0f);
    f.rh = fmaxf(ye - f.ys, 0.0f);
  } else {
    f.xs = roi[1] * scale - 0.5f;
    f.ys = roi[2] * scale - 0.5f;
    const float xe = roi[3] * scale - 0.5f, ye = roi[4] * scale - 0.5f;
    f.rw = xe - f.xs;
    f.rh = ye - f.ys;
  }
  f.bin_h = f.rh / (float)ph;
  f.bin_w = f.rw / (float)pw;
  f.gh = grid_of(f.bin_h, sample_num);
  f.gw = grid_of(f.bin_w, sample_num);
  const long long ci = (long long)f.gh * f.gw;
  f.cnt = (kind == 1) ? (float)ci : (float)(ci < 1 ? 1 : ci);
}

__device__ __forceinline__ void sample_xy(const RoiFrame &f, int ph_i, int pw_i, int iy, int ix,
                                          float &x, float &y) {
  if (f.kind == 0) {
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
  } else {   // roi_start + p * bin + (i + .5) * bin / grid, left to right (roi_align_kernel.cu:100-107)
    float ty = f.ys + (float)ph_i * f.bin_h;
    float t1 = ((float)iy + 0.5f) * f.bin_h;
    y = ty + t1 / (float)f.gh;
    float tx = f.xs + (float)pw_i * f.bin_w;
    float t2 = ((float)ix + 0.5f) * f.bin_w;
    x = tx + t2 / (float)f.gw;
  }
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

// same rules, but returning the tap coordinates (no division to recover them)
__device__ __forceinline__ bool bilinear_taps_xy(int height, int width, float y, float x, int &yl,
                                                 int &xl, int &yh, int &xh, float (&w)[4]) {
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
  yl = y_low; xl = x_low; yh = y_high; xh = x_high;
  w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
  if (outside) { w[0] = w[1] = w[2] = w[3] = 0.0f; }
  return !outside;
}

struct Footprint { int x0, y0, w, h; };

// feature pyramid of the multi-level forward (obb_roi_align_levels_forward_f32): passed by value
struct LevelTable {
  const float *feat[OBB_MAX_LEVELS];
  int H[OBB_MAX_LEVELS], W[OBB_MAX_LEVELS];
  float scale[OBB_MAX_LEVELS];
};

// The axis-aligned pixel box that holds every bilinear tap of every sample.  The samples
// are an affine lattice, so their extreme coordinates are those of the four CORNER samples,
// evaluated by the very expressions the sample table uses; interior samples repeat those
// expressions with other small integers, i.e. they may exceed a corner by a few ulps of the
// coordinate: margin 1e-3 px + 1e-6 |coordinate| (>= 8 ulps).  (The rectangle's own corners
// plus a pixel of slack -- the first version -- cost ~3 more rows and columns per RoI: a
// third of the bytes staged at configs[2].)
__device__ __forceinline__ Footprint roi_footprint(const RoiFrame &f, int H, int W, int ph, int pw) {
  float lox, hix, loy, hiy;
  {
    float x, y;
    sample_xy(f, 0, 0, 0, 0, x, y);
    lox = hix = x; loy = hiy = y;
    sample_xy(f, 0, pw - 1, 0, f.gw - 1, x, y);
    lox = min_sel(lox, x); hix = max_sel(hix, x); loy = min_sel(loy, y); hiy = max_sel(hiy, y);
    sample_xy(f, ph - 1, 0, f.gh - 1, 0, x, y);
    lox = min_sel(lox, x); hix = max_sel(hix, x); loy = min_sel(loy, y); hiy = max_sel(hiy, y);
    sample_xy(f, ph - 1, pw - 1, f.gh - 1, f.gw - 1, x, y);
    lox = min_sel(lox, x); hix = max_sel(hix, x); loy = min_sel(loy, y); hiy = max_sel(hiy, y);
  }
  const float mx = 1.0e-3f + 1.0e-6f * max_sel(fabs_sel(lox), fabs_sel(hix));
  const float my = 1.0e-3f + 1.0e-6f * max_sel(fabs_sel(loy), fabs_sel(hiy));
  // low tap of the smallest coordinate .. high tap (+ 1) of the largest
  float fx0 = floorf(lox - mx), fx1 = floorf(hix + mx) + 1.0f;
  float fy0 = floorf(loy - my), fy1 = floorf(hiy + my) + 1.0f;
  fx0 = min_sel(max_sel(fx0, 0.0f), (float)(W - 1)); fx1 = min_sel(max_sel(fx1, 0.0f), (float)(W - 1));
  fy0 = min_sel(max_sel(fy0, 0.0f), (float)(H - 1)); fy1 = min_sel(max_sel(fy1, 0.0f), (float)(H - 1));
  Footprint p;
  p.x0 = (int)fx0; p.y0 = (int)fy0;
  p.w = (int)fx1 - p.x0 + 1; p.h = (int)fy1 - p.y0 + 1;
  return p;
}

// The taps of one bin that land on the SAME pixel become one entry: the first tap of such a group is its leader and
// carries the sum of the group's weights (ascending tap number: a fixed order).  A bin's taps are `tpb` consecutive tap
// numbers = consecutive lanes (tpb a power of two <= 64, the lane groups aligned).  pix < 0: no tap in this lane.
__device__ __forceinline__ void merge_bin_taps(int pix, float w, int lane, int tpb, bool &lead, float &sum_out) {
  const int g0 = lane & ~(tpb - 1), me = lane - g0;
  float sum = 0.0f;
  bool first = true;
  if (tpb == 16) {
    // 2 x 2 samples per bin, the usual case: a bin's taps are one DPP row.  Rotating the row by 16 - d brings lane
    // me + d (mod 16): d = 1 .. 15 visits me + 1 .. 15 (summed, ascending: the order of the loop below), then
    // 0 .. me - 1 (the leader test).  VALU moves only -- the same exchanges through ds_bpermute (128 per thread)
    // kept the LDS pipe of the CU busy for 4.8 us of a 12.6 us plan block.
    sum = pix >= 0 ? w : 0.0f;
#define OBB_ROW_STEP(D)                                                                                       \
    {                                                                                                         \
      const int pj = __builtin_amdgcn_update_dpp(0, pix, 0x120 + (16 - (D)), 0xF, 0xF, false);                \
      const float wj = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w), 0x120 + (16 - (D)), 0xF, 0xF, false)); \
      if (pix >= 0 && pj == pix) { if (me + (D) >= 16) first = false; else sum += wj; }                        \
    }
    OBB_ROW_STEP(1) OBB_ROW_STEP(2) OBB_ROW_STEP(3) OBB_ROW_STEP(4) OBB_ROW_STEP(5) OBB_ROW_STEP(6) OBB_ROW_STEP(7)
    OBB_ROW_STEP(8) OBB_ROW_STEP(9) OBB_ROW_STEP(10) OBB_ROW_STEP(11) OBB_ROW_STEP(12) OBB_ROW_STEP(13)
    OBB_ROW_STEP(14) OBB_ROW_STEP(15)
#undef OBB_ROW_STEP
  } else {
    for (int j = 0; j < tpb; ++j) {                      // ascending tap number: a fixed order
      const int pj = __shfl(pix, g0 + j);
      const float wj = __shfl(w, g0 + j);
      if (pix >= 0 && pj == pix) { if (j < me) first = false; else sum += wj; }
    }
  }
  lead = pix >= 0 && first;
  sum_out = sum;
}

// ---- the per-RoI plan of the gather backward (rroi_plan_kernel, obb_rroi.hip) ------------------------------------------
constexpr int kMaxTabSamples = 256;   // sample table (8 KiB); 7x7 bins x 2x2 samples = 196
constexpr int kBwdMaxPix = 2047;                       // footprint pixels of a planned RoI
constexpr int kBwdMaxEntries = 4 * kMaxTabSamples;     // taps of a RoI (4 per sample)
constexpr int kPlanPix = kBwdMaxPix + 2;               // CSR offsets per RoI (16 bit)
constexpr int kLptMax = 4096;                          // map tiles (all images) whose order is worked out: tile counts | ticket | order
// plan record; state 1: planned (rstart / rentry hold "footprint pixel -> (bin, weight / count) of every tap on it", the
// taps of a pixel in tap order), 2: left to an atomics kernel, 0: nothing sampled
struct alignas(16) RoiPlan { int b, x0, y0, w, h, state, nb, pad; };
struct GatherLayout { size_t plan, rstart, rentry, lpt, total; };
GatherLayout gather_layout(int64_t K);
int rroi_plan_launch(int B, int H, int W, const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale, int sample_num,
                     float min_size, int roi_kind, RoiPlan *plan, unsigned short *rstart, int2 *rentry, int *lpt, int tiles_x,
                     int tiles, void *stream, const float *tg = nullptr, float *tgT = nullptr, int tC = 0);

// host helpers of obb_rroi.hip used by obb_rroi_nhwc.hip
constexpr int kOrderMaxK = 65536;    // above: identity order (the caller passes no order)
int rroi_check(const char *who, int B, int C, int H, int W, int64_t K, int ph, int pw);
// order[rank] = RoI: size class | pyramid level | Morton code of the centre (rroi_order_kernel); order_ws holds 2 K ints
int rroi_order_launch(const LevelTable &lv, const int *roi_level_dev, int num_levels, const float *rois_dev, int64_t K,
                      int roi_kind, int *order_ws, void *stream);

}  // namespace obb
