// obb_rroi.hip -- rotated (and horizontal) RoI-Align forward / backward for gfx950.
//
// Rotated counterpart of ROIAlignForwardV1 / RoIAlignForwardV2
// (mmdet/ops/roi_align/src/roi_align_kernel.cu:64-117, roi_align_kernel_v2.cu:62-128)
// and RoIAlignBackwardFeatureV2 (roi_align_kernel_v2.cu:179-263), plus the horizontal ops
// themselves in their own expression order (the reference's .cu files need THC headers and
// cannot be built on ROCm).  Same bilinear tap rules (roi_align_kernel_v2.cu:15-60); for a
// rotated RoI the sampling grid is laid out in the RoI's own frame and rotated by theta
// about the centre.
//
// Design.  The reference gives every OUTPUT ELEMENT a thread that recomputes coordinates and
// weights and gathers 16 scattered floats from HBM/L2.  Here a workgroup owns (RoI, channel
// chunk):
//   1. the sample table (4 tap offsets + 4 weights per sample) is computed ONCE per RoI and
//      block into LDS -- it is the same for every channel;
//   2. the RoI's footprint (the axis-aligned pixel box that holds every tap) of a group of CB
//      channel planes is staged into LDS TRANSPOSED, as tile[pixel][channel] with an odd
//      pitch CB + 1: global reads run along x (row segments, coalesced), LDS writes of
//      consecutive pixels land in consecutive banks;
//   3. a lane is (bin, channel) with the CHANNEL in the low lane bits: the 64 / CB bins a
//      wave works on read 64 / CB pixels, and the CB lanes of a bin read CB consecutive
//      words -- bank-conflict free by layout, every lane busy (the previous kernel had a lane
//      per bin: 49 of 64 lanes, ~3.5-way conflicts on the random gathers);
//   4. results go through a small LDS tile back into (channel, bin) order and leave as
//      contiguous runs (the output of a block is one contiguous range).
// Values and operation order per output are those of the oracle (sum over samples of
// w1*v1 + w2*v2 + w3*v3 + w4*v4, then / count).
//
// L2.  Workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2.  Block
// ids are laid out so that at any time one XCD works on ONE channel chunk (for all RoIs), and
// the chunk is sized so that its planes fit that L2 (8 channels of a 256 x 256 map = 2 MiB):
// the map is then fetched from HBM about once instead of once per RoI that touches it.
#include <cstdlib>

#include "obb_common.h"
#include "obb_geom.cuh"

namespace obb {

constexpr int kChanPerThread = 8;   // channel grouping of the direct (unstaged) paths

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

// ---- forward, direct gathers (any footprint size) ---------------------------------------
// item = (channel group, bin); used for RoIs whose footprint does not fit the LDS tile
__device__ __forceinline__ void rroi_forward_direct(const float *__restrict__ feat, int C, int H,
                                                    int W, const RoiFrame &f, int64_t k, int ph,
                                                    int pw, int c_block, int c_end,
                                                    float *__restrict__ out) {
  const int nb = ph * pw;
  const int64_t plane = (int64_t)H * W;
  const int groups = (c_end - c_block + kChanPerThread - 1) / kChanPerThread;
  for (int it = threadIdx.x; it < groups * nb; it += 256) {
    const int grp = it / nb, bin = it - grp * nb;
    const int c0 = c_block + grp * kChanPerThread;
    const int bi = bin / pw, bj = bin - bi * pw;
    float acc[kChanPerThread];
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c) acc[c] = 0.0f;
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
          if (c0 + c < c_end) {
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
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c)
      if (c0 + c < c_end) out[((k * C + c0 + c) * (int64_t)nb) + bin] = acc[c] / f.cnt;
  }
}

// ---- staged kernels: shared pieces ---------------------------------------------------------
#ifndef OBB_STAGE_FLOATS
#define OBB_STAGE_FLOATS 6144
#endif
constexpr int kStageFloats = OBB_STAGE_FLOATS;  // input tile [pixel][channel + 1] (24 KiB)
constexpr int kMaxTabSamples = 256;   // sample table (8 KiB); 7x7 bins x 2x2 samples = 196
constexpr int kMaxBins = 64;          // bins of a staged RoI (output tile 64 x 65 floats)
constexpr int kBinPitchMax = 65;

struct Footprint { int x0, y0, w, h; };

__device__ __forceinline__ Footprint roi_footprint(const RoiFrame &f, int H, int W) {
  float lox, hix, loy, hiy;
  if (f.kind == 0) {
    const float hx = 0.5f * (fabs_sel(f.c) * f.rw + fabs_sel(f.s) * f.rh);
    const float hy = 0.5f * (fabs_sel(f.s) * f.rw + fabs_sel(f.c) * f.rh);
    lox = f.cx - hx; hix = f.cx + hx; loy = f.cy - hy; hiy = f.cy + hy;
  } else {
    const float ex = f.xs + f.rw, ey = f.ys + f.rh;
    lox = min_sel(f.xs, ex); hix = max_sel(f.xs, ex);
    loy = min_sel(f.ys, ey); hiy = max_sel(f.ys, ey);
  }
  // one pixel of slack on each side covers the high tap and any rounding
  float fx0 = floorf(lox) - 1.0f, fx1 = floorf(hix) + 2.0f;
  float fy0 = floorf(loy) - 1.0f, fy1 = floorf(hiy) + 2.0f;
  fx0 = min_sel(max_sel(fx0, 0.0f), (float)(W - 1)); fx1 = min_sel(max_sel(fx1, 0.0f), (float)(W - 1));
  fy0 = min_sel(max_sel(fy0, 0.0f), (float)(H - 1)); fy1 = min_sel(max_sel(fy1, 0.0f), (float)(H - 1));
  Footprint p;
  p.x0 = (int)fx0; p.y0 = (int)fy0;
  p.w = (int)fx1 - p.x0 + 1; p.h = (int)fy1 - p.y0 + 1;
  return p;
}

// XCD-aware block -> (RoI, channel chunk): consecutive ids take the SAME RoI and 8 DIFFERENT
// chunks (one per XCD under round-robin dispatch); chunks 8r .. 8r+7 ("round" r) are swept
// over all RoIs before round r + 1 starts.
__device__ __forceinline__ bool block_to_work(int K, int chunks, int64_t &k, int &chunk) {
  const uint32_t lin = blockIdx.x;
  const uint32_t c8 = lin & 7u, rest = lin >> 3;
  k = rest % (uint32_t)K;
  chunk = (int)((rest / (uint32_t)K) * 8u + c8);
  return chunk < chunks;
}

// channels per staging pass: the largest power of two <= 64 whose tile fits, not (much) more
// than the block owns
__device__ __forceinline__ int pass_channels(int fsz, int owned) {
  int cb = 64;
  while (cb > 1 && (fsz * (cb + 1) > kStageFloats || (cb >> 1) >= owned)) cb >>= 1;
  return cb;
}

// sample table: tap offsets (pixel index in the footprint x pitch) and weights, once per block
__device__ __forceinline__ void build_table(const RoiFrame &f, const Footprint &fp, int H, int W, int pw,
                                            int spb, int nsamp, int pitch, int4 *s_off, float4 *s_wgt) {
  for (int sidx = threadIdx.x; sidx < nsamp; sidx += 256) {
    const int bin = sidx / spb, q = sidx - bin * spb;
    const int bi = bin / pw, bj = bin - bi * pw;
    const int iy = q / f.gw, ix = q - iy * f.gw;
    float x, y, w[4];
    int yl, xl, yh, xh;
    sample_xy(f, bi, bj, iy, ix, x, y);
    const bool live = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
    // tap coordinates -> footprint-local pixel index (clamped: never out of the tile)
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    const int ly0 = cl(yl - fp.y0, fp.h) * fp.w, ly1 = cl(yh - fp.y0, fp.h) * fp.w;
    const int lx0 = cl(xl - fp.x0, fp.w), lx1 = cl(xh - fp.x0, fp.w);
    s_off[sidx] = live ? make_int4((ly0 + lx0) * pitch, (ly0 + lx1) * pitch, (ly1 + lx0) * pitch, (ly1 + lx1) * pitch)
                       : make_int4(0, 0, 0, 0);
    s_wgt[sidx] = make_float4(w[0], w[1], w[2], w[3]);
  }
}

// ---- forward, staged -----------------------------------------------------------------------
// grid: (K x channel chunks, XCD-aware order); block: 256 threads.
__global__ __launch_bounds__(256) void rroi_forward_kernel(
    const float *__restrict__ feat, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int chan_per_block,
    int K, int chunks, int roi_kind, float *__restrict__ out) {
  __shared__ float tile[kStageFloats];
  __shared__ int4 s_off[kMaxTabSamples];
  __shared__ float4 s_wgt[kMaxTabSamples];
  __shared__ float otile[64 * kBinPitchMax];
  int64_t k;
  int chunk;
  if (!block_to_work(K, chunks, k, chunk)) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  const int nb = ph * pw;
  const int c_block = chunk * chan_per_block;
  const int c_end = min(C, c_block + chan_per_block);
  const int64_t plane = (int64_t)H * W;
  float *oblk = out + (k * C + c_block) * (int64_t)nb;       // this block's outputs are contiguous
  const int nout = (c_end - c_block) * nb;
  const long long spb64 = (long long)f.gh * f.gw;
  if (f.b < 0 || f.b >= B || spb64 <= 0) {
    // invalid batch index: zeros.  Empty sampling grid: the sum of nothing over the count --
    // 0 (V2, rotated) or 0/0 = NaN (V1 divides by gh*gw, roi_align_kernel.cu:113)
    const float v = (f.b < 0 || f.b >= B) ? 0.0f : 0.0f / f.cnt;
    for (int it = threadIdx.x; it < nout; it += 256) oblk[it] = v;
    return;
  }
  const Footprint fp = roi_footprint(f, H, W);
  const int fsz = fp.w * fp.h;                       // pixels per channel plane
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : spb64);   // samples per bin
  const int nsamp = nb * spb;
  // block-uniform: huge RoI / huge sampling grid / >2^31-element tensors gather directly
  if (2 * fsz > kStageFloats || nb > kMaxBins || nsamp > kMaxTabSamples || (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_forward_direct(feat, C, H, W, f, k, ph, pw, c_block, c_end, out);
    return;
  }
  const int cb = pass_channels(fsz, c_end - c_block);  // channels per pass (power of two)
  const int pitch = cb + 1;
  build_table(f, fp, H, W, pw, spb, nsamp, pitch, s_off, s_wgt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & (cb - 1);                    // channel of this lane inside the pass
  const int bsub = lane / cb, bpw = 64 / cb;         // bin slot of this lane, bins per wave step
  const int nbp = nb | 1;                            // odd pitch of the output tile
  const uint32_t m_fsz = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;   // exact idx / fsz for idx < 2^16
  const uint32_t m_w = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
  const uint32_t m_nb = nb > 1 ? 0xFFFFFFFFu / (uint32_t)nb + 1u : 0u;
  for (int cp = c_block; cp < c_end; cp += cb) {
    const int nch = min(cb, c_end - cp);
    // stage nch footprints, transposed: consecutive threads read consecutive x of a row
    const float *src = feat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    const int total = nch * fsz;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      const uint32_t c = fsz > 1 ? __umulhi((uint32_t)idx, m_fsz) : (uint32_t)idx;
      const uint32_t p = (uint32_t)idx - c * (uint32_t)fsz;
      const uint32_t y = fp.w > 1 ? __umulhi(p, m_w) : p;
      const uint32_t x = p - y * (uint32_t)fp.w;
      tile[p * (uint32_t)pitch + c] = src[(int64_t)c * plane + (int64_t)y * W + x];
    }
    __syncthreads();   // tile (and, in the first pass, the table) complete
    for (int bin0 = wave * bpw; bin0 < nb; bin0 += 4 * bpw) {
      const int bin = bin0 + bsub;
      const bool on = bin < nb && lc < nch;
      const int tb = (bin < nb ? bin : nb - 1) * spb;
      const float *p = tile + lc;
      float acc = 0.0f;
      for (int q = 0; q < spb; ++q) {
        const int4 o = s_off[tb + q];
        const float4 w = s_wgt[tb + q];
        float v = w.x * p[o.x];
        v = v + w.y * p[o.y];
        v = v + w.z * p[o.z];
        v = v + w.w * p[o.w];
        acc = acc + v;
      }
      if (on) otile[lc * nbp + bin] = acc / f.cnt;
    }
    __syncthreads();
    // (channel, bin) order = the output's own order: one contiguous run per pass
    float *dst = out + (k * C + cp) * (int64_t)nb;
    for (int i = threadIdx.x; i < nch * nb; i += 256) {
      const int c = nb > 1 ? (int)__umulhi((uint32_t)i, m_nb) : i, bin = i - c * nb;
      dst[i] = otile[c * nbp + bin];
    }
    // no barrier here: the next pass writes `tile` (every thread is past the compute loop) and
    // touches `otile` only after its own first barrier (every thread is past this loop)
  }
}

// ---- backward, direct atomics (any footprint size) -----------------------------------------
__device__ __forceinline__ void rroi_backward_direct(const float *__restrict__ gout, int C, int H, int W,
                                                     const RoiFrame &f, int64_t k, int ph, int pw,
                                                     int c_block, int c_end,
                                                     float *__restrict__ gfeat) {
  const int nb = ph * pw;
  const int groups = (c_end - c_block + kChanPerThread - 1) / kChanPerThread;
  const int64_t plane = (int64_t)H * W;
  for (int it = threadIdx.x; it < groups * nb; it += 256) {
    const int grp = it / nb, bin = it - grp * nb;
    const int c0 = c_block + grp * kChanPerThread;
    const int bi = bin / pw, bj = bin - bi * pw;
    float g[kChanPerThread];
#pragma unroll
    for (int c = 0; c < kChanPerThread; ++c)
      g[c] = (c0 + c < c_end) ? gout[((k * C + c0 + c) * (int64_t)nb) + bin] : 0.0f;
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
          if (c0 + c < c_end) {
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

// ---- backward, staged -----------------------------------------------------------------------
// Mirror image of the forward kernel: the gradient of a pass's channels is accumulated in the
// transposed LDS tile (ds_add_f32; lane = (bin, channel), so the lanes of a bin add to
// consecutive banks and taps that neighbouring bins share meet in LDS, not in L2), then the
// touched pixels are added to the gradient map with global float atomics, in row segments of
// consecutive addresses.  Float atomics commute only up to rounding: tolerance 1e-4 * max|g|.
__global__ __launch_bounds__(256) void rroi_backward_kernel(
    const float *__restrict__ gout, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int chan_per_block,
    int K, int chunks, int roi_kind, float *__restrict__ gfeat) {
  __shared__ float tile[kStageFloats];
  __shared__ int4 s_off[kMaxTabSamples];
  __shared__ float4 s_wgt[kMaxTabSamples];
  __shared__ float gtile[64 * kBinPitchMax];
  int64_t k;
  int chunk;
  if (!block_to_work(K, chunks, k, chunk)) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  const long long spb64 = (long long)f.gh * f.gw;
  if (f.b < 0 || f.b >= B || spb64 <= 0) return;     // nothing was sampled: no gradient
  const int nb = ph * pw;
  const int c_block = chunk * chan_per_block;
  const int c_end = min(C, c_block + chan_per_block);
  const int64_t plane = (int64_t)H * W;
  const Footprint fp = roi_footprint(f, H, W);
  const int fsz = fp.w * fp.h;
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : spb64);
  const int nsamp = nb * spb;
  if (2 * fsz > kStageFloats || nb > kMaxBins || nsamp > kMaxTabSamples || (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_backward_direct(gout, C, H, W, f, k, ph, pw, c_block, c_end, gfeat);
    return;
  }
  const int cb = pass_channels(fsz, c_end - c_block);
  const int pitch = cb + 1;
  build_table(f, fp, H, W, pw, spb, nsamp, pitch, s_off, s_wgt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane & (cb - 1);
  const int bsub = lane / cb, bpw = 64 / cb;
  const int nbp = nb | 1;
  const uint32_t m_fsz = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;
  const uint32_t m_w = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
  const uint32_t m_nb = nb > 1 ? 0xFFFFFFFFu / (uint32_t)nb + 1u : 0u;
  for (int cp = c_block; cp < c_end; cp += cb) {
    const int nch = min(cb, c_end - cp);
    for (int i = threadIdx.x; i < fsz * pitch; i += 256) tile[i] = 0.0f;
    const float *gsrc = gout + (k * C + cp) * (int64_t)nb;   // contiguous (channel, bin) run
    for (int i = threadIdx.x; i < nch * nb; i += 256) {
      const int c = nb > 1 ? (int)__umulhi((uint32_t)i, m_nb) : i, bin = i - c * nb;
      gtile[c * nbp + bin] = gsrc[i];
    }
    __syncthreads();
    for (int bin0 = wave * bpw; bin0 < nb; bin0 += 4 * bpw) {
      const int bin = bin0 + bsub;
      if (bin < nb && lc < nch) {
        const float g = gtile[lc * nbp + bin];
        float *p = tile + lc;
        for (int q = 0; q < spb; ++q) {
          const int4 o = s_off[bin * spb + q];
          const float4 w = s_wgt[bin * spb + q];
          // same per-tap arithmetic as the direct kernel: (g * w) / count
          atomicAdd(p + o.x, (g * w.x) / f.cnt);
          atomicAdd(p + o.y, (g * w.y) / f.cnt);
          atomicAdd(p + o.z, (g * w.z) / f.cnt);
          atomicAdd(p + o.w, (g * w.w) / f.cnt);
        }
      }
    }
    __syncthreads();
    float *dst = gfeat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    const int total = nch * fsz;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      const uint32_t c = fsz > 1 ? __umulhi((uint32_t)idx, m_fsz) : (uint32_t)idx;
      const uint32_t p = (uint32_t)idx - c * (uint32_t)fsz;
      const float v = tile[p * (uint32_t)pitch + c];
      if (v != 0.0f) {
        const uint32_t y = fp.w > 1 ? __umulhi(p, m_w) : p;
        const uint32_t x = p - y * (uint32_t)fp.w;
        atomicAdd(dst + (int64_t)c * plane + (int64_t)y * W + x, v);
      }
    }
    __syncthreads();   // the next pass clears the tile
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

}  // namespace obb

namespace obb {

// channels per block: the planes one XCD works on at a time (one chunk, see block_to_work)
// should fit its 4 MiB L2 with room to spare -- ~2 MiB of planes -- within [8, 64]
static int rroi_chan_per_block(int C, int H, int W) {
  const long long plane_bytes = (long long)H * W * 4;
  int cpb = 64;
  while (cpb > 8 && cpb * plane_bytes > (2ll << 20)) cpb >>= 1;
  const int64_t g = tuning(OBB_TUNE_RROI_GROUPS, -1);   // test / experiment hook: groups of 8 channels
  if (g > 0 && g <= 64) cpb = (int)g * kChanPerThread;
  (void)C;
  return cpb;
}

static int rroi_geometry(const char *who, int C, int H, int W, int64_t K, int &cpb, unsigned &chunks,
                         unsigned long long &nblk) {
  cpb = rroi_chan_per_block(C, H, W);
  chunks = (unsigned)((C + cpb - 1) / cpb);
  nblk = (unsigned long long)((chunks + 7u) / 8u) * 8ull * (unsigned long long)K;
  if (nblk > 2147483647ull) return set_err(OBB_EINVAL, "%s: too many blocks (K=%lld, C=%d)", who, (long long)K, C);
  return OBB_OK;
}

static int rroi_forward_launch(const char *who, const float *feat_dev, int B, int C, int H, int W,
                               const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale,
                               int sample_num, float min_size, int roi_kind, float *out_dev, void *stream) {
  int rc = rroi_check(who, B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (K == 0 || C == 0) return OBB_OK;
  if (!feat_dev || !rois_dev || !out_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (H == 0 || W == 0 || B == 0) return set_err(OBB_EINVAL, "%s: empty feature map with %lld RoIs", who, (long long)K);
  int cpb;
  unsigned chunks;
  unsigned long long nblk;
  rc = rroi_geometry(who, C, H, W, K, cpb, chunks, nblk);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(rroi_forward_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     feat_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, cpb, (int)K, (int)chunks, roi_kind, out_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

static int rroi_backward_launch(const char *who, const float *grad_out_dev, int B, int C, int H, int W,
                                const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale,
                                int sample_num, float min_size, int roi_kind, float *grad_feat_dev,
                                void *stream) {
  int rc = rroi_check(who, B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (K == 0 || C == 0 || H == 0 || W == 0 || B == 0) return OBB_OK;
  if (!grad_out_dev || !rois_dev || !grad_feat_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  int cpb;
  unsigned chunks;
  unsigned long long nblk;
  rc = rroi_geometry(who, C, H, W, K, cpb, chunks, nblk);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(rroi_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     grad_out_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, cpb, (int)K, (int)chunks, roi_kind, grad_feat_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

}  // namespace obb

extern "C" int obb_rroi_align_forward_f32(const float *feat_dev, int B, int C, int H, int W,
                                          const float *rois_dev, int64_t K, int ph, int pw,
                                          float spatial_scale, int sample_num, float min_size,
                                          float *out_dev, void *stream) {
  return obb::rroi_forward_launch("obb_rroi_align_forward_f32", feat_dev, B, C, H, W, rois_dev, K, ph, pw,
                                  spatial_scale, sample_num, min_size, 0, out_dev, stream);
}

extern "C" int obb_rroi_align_backward_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                           const float *rois_dev, int64_t K, int ph, int pw,
                                           float spatial_scale, int sample_num, float min_size,
                                           float *grad_feat_dev, void *stream) {
  return obb::rroi_backward_launch("obb_rroi_align_backward_f32", grad_out_dev, B, C, H, W, rois_dev, K, ph, pw,
                                   spatial_scale, sample_num, min_size, 0, grad_feat_dev, stream);
}

extern "C" int obb_roi_align_forward_f32(const float *feat_dev, int B, int C, int H, int W,
                                         const float *rois_dev, int64_t K, int ph, int pw,
                                         float spatial_scale, int sample_num, int aligned,
                                         float *out_dev, void *stream) {
  return obb::rroi_forward_launch("obb_roi_align_forward_f32", feat_dev, B, C, H, W, rois_dev, K, ph, pw,
                                  spatial_scale, sample_num, 0.0f, aligned ? 2 : 1, out_dev, stream);
}

extern "C" int obb_roi_align_backward_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                          const float *rois_dev, int64_t K, int ph, int pw,
                                          float spatial_scale, int sample_num, int aligned,
                                          float *grad_feat_dev, void *stream) {
  return obb::rroi_backward_launch("obb_roi_align_backward_f32", grad_out_dev, B, C, H, W, rois_dev, K, ph, pw,
                                   spatial_scale, sample_num, 0.0f, aligned ? 2 : 1, grad_feat_dev, stream);
}
