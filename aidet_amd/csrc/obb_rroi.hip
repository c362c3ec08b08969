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
#include <cstdlib>

#include "obb_common.h"
#include "obb_geom.cuh"

namespace obb {

constexpr int kChanPerThread = 8;

struct RoiFrame {
  int b, gh, gw;
  float cx, cy, c, s, bin_w, bin_h, x0, y0, cnt;
};

// RoI row layouts.  kind 0: [b, cx, cy, w, h, theta].  The horizontal RoIs of the reference's
// RoIAlign, [b, x1, y1, x2, y2] (roi_align.py:136), are the theta = 0 special cases
//   kind 1 (legacy V1, roi_align_kernel.cu:79-86): centre ((x1+x2+1)/2, (y1+y2+1)/2), size
//          (x2-x1+1, y2-y1+1), no lower clamp
//   kind 2 (V2 aligned=True, roi_align_kernel_v2.cu:79-86): centre ((x1+x2)/2 - off,
//          (y1+y2)/2 - off) with off = 0.5 / spatial_scale, size (x2-x1, y2-y1), no clamp
// evaluated here in the same fp32 operations the former torch-side mapping used.
__device__ __forceinline__ int roi_stride(int kind) { return kind == 0 ? 6 : 5; }

__device__ __forceinline__ void make_frame(const float *__restrict__ roi, int kind, float scale, int ph,
                                           int pw, int sample_num, float min_size, RoiFrame &f) {
  f.b = (int)roi[0];
  float cx, cy, w, h, theta;
  if (kind == 0) {
    cx = roi[1]; cy = roi[2]; w = roi[3]; h = roi[4]; theta = roi[5];
  } else {
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    theta = 0.0f;
    min_size = 0.0f;
    if (kind == 1) {
      cx = ((x1 + x2) + 1.0f) * 0.5f; cy = ((y1 + y2) + 1.0f) * 0.5f;
      w = (x2 - x1) + 1.0f; h = (y2 - y1) + 1.0f;
    } else {
      const float off = 0.5f / scale;
      cx = (x1 + x2) * 0.5f - off; cy = (y1 + y2) * 0.5f - off;
      w = x2 - x1; h = y2 - y1;
    }
  }
  f.cx = cx * scale;
  f.cy = cy * scale;
  float rw = w * scale, rh = h * scale;
  rw = max_sel(rw, min_size);
  rh = max_sel(rh, min_size);
  sincos_det(theta, f.s, f.c);
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

// ---- forward, staged ---------------------------------------------------------------
// grid: (K, channel chunks); block: 256 threads; item = (channel group, bin).
// The RoI's footprint (the axis-aligned pixel box that contains every bilinear tap
// of every sample) of each channel plane is first copied into LDS with coalesced row
// loads; the ~4 x samples x channels taps are then LDS gathers instead of scattered
// 4-byte global loads.  Values and operation order are unchanged.
#ifndef OBB_STAGE_FLOATS
#define OBB_STAGE_FLOATS 6144
#endif
constexpr int kStageFloats = OBB_STAGE_FLOATS;  // LDS tile per block (24 KiB)
constexpr int kWaveFloats = kStageFloats / 4;   // private quarter of a wave (small footprints)
constexpr int kWavePad = 256;                   // room for the tail lanes of a wave's last DMA instruction
constexpr int kMaxTabSamples = 256;   // sample table (8 KiB); 7x7 bins x 2x2 samples = 196

struct Footprint { int x0, y0, w, h; };

__device__ __forceinline__ Footprint roi_footprint(const RoiFrame &f, int ph, int pw, int H, int W) {
  const float rw = f.bin_w * (float)pw, rh = f.bin_h * (float)ph;
  const float hx = 0.5f * (fabs_sel(f.c) * rw + fabs_sel(f.s) * rh);
  const float hy = 0.5f * (fabs_sel(f.s) * rw + fabs_sel(f.c) * rh);
  // one pixel of slack on each side covers the high tap and any rounding
  float fx0 = floorf(f.cx - hx) - 1.0f, fx1 = floorf(f.cx + hx) + 2.0f;
  float fy0 = floorf(f.cy - hy) - 1.0f, fy1 = floorf(f.cy + hy) + 2.0f;
  fx0 = min_sel(max_sel(fx0, 0.0f), (float)(W - 1)); fx1 = min_sel(max_sel(fx1, 0.0f), (float)(W - 1));
  fy0 = min_sel(max_sel(fy0, 0.0f), (float)(H - 1)); fy1 = min_sel(max_sel(fy1, 0.0f), (float)(H - 1));
  Footprint p;
  p.x0 = (int)fx0; p.y0 = (int)fy0;
  p.w = (int)fx1 - p.x0 + 1; p.h = (int)fy1 - p.y0 + 1;
  return p;
}

template <bool X4>
__global__ __launch_bounds__(256) void rroi_forward_kernel(
    const float *__restrict__ feat, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int groups_per_block,
    int K, int chunks, int roi_kind, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[4 * (kWaveFloats + kWavePad)];  // block-wide: kStageFloats + pad
  __shared__ int4 s_off[kMaxTabSamples];      // per sample: 4 footprint-local tap offsets
  __shared__ float4 s_wgt[kMaxTabSamples];    // per sample: 4 bilinear weights
  // XCD-aware block -> (RoI, channel chunk): workgroups are dealt round-robin over the 8
  // XCDs (each with its own 4 MiB L2), so consecutive ids take the SAME RoI and 8
  // DIFFERENT channel chunks: every XCD then sweeps all RoIs over its own chunk of
  // planes, which stays L2 resident, instead of all 8 L2s thrashing on every plane.
  const uint32_t lin = blockIdx.x;
  const uint32_t c8 = lin & 7u, rest = lin >> 3;
  const int64_t k = rest % (uint32_t)K;
  const int chunk = (int)((rest / (uint32_t)K) * 8u + c8);
  if (chunk >= chunks) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  const int nb = ph * pw;
  const int c_block = chunk * groups_per_block * kChanPerThread;
  const int c_end = min(C, c_block + groups_per_block * kChanPerThread);
  const bool bad = f.b < 0 || f.b >= B;
  const int64_t plane = (int64_t)H * W;
  if (bad) {  // zeros for an invalid batch index
    for (int it = threadIdx.x; it < (c_end - c_block) * nb; it += 256)
      out[(k * C + c_block) * (int64_t)nb + it] = 0.0f;
    return;
  }
  Footprint fp = roi_footprint(f, ph, pw, H, W);
  if (X4) {  // 16-byte LDS-DMA: rows start and end on multiples of 4 floats (W % 4 == 0)
    const int x1 = (fp.x0 + fp.w - 1) | 3;
    fp.x0 &= ~3;
    fp.w = x1 - fp.x0 + 1;
  }
  const int fsz = fp.w * fp.h;                       // floats per channel plane
  const int spb = f.gh * f.gw;                       // samples per bin
  const int nsamp = nb * spb;
  // block-uniform: huge RoI / huge sampling grid / >2^31-element tensors gather directly
  if (fsz > kStageFloats || nsamp > kMaxTabSamples || (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_forward_direct(feat, C, H, W, f, k, ph, pw, c_block, c_end, out);
    return;
  }
  // sample table: coordinates, taps and weights are computed ONCE per RoI and block
  // (the reference recomputes them for every output element, roi_align_kernel.cu:70-113)
  for (int sidx = threadIdx.x; sidx < nsamp; sidx += 256) {
    const int bin = sidx / spb, q = sidx - bin * spb;
    const int bi = bin / pw, bj = bin - bi * pw;
    const int iy = q / f.gw, ix = q - iy * f.gw;
    float x, y, w[4];
    int yl, xl, yh, xh;
    sample_xy(f, bi, bj, iy, ix, x, y);
    const bool live = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
    // tap coordinates -> footprint-local offsets (clamped: never out of the tile)
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    const int ly0 = cl(yl - fp.y0, fp.h) * fp.w, ly1 = cl(yh - fp.y0, fp.h) * fp.w;
    const int lx0 = cl(xl - fp.x0, fp.w), lx1 = cl(xh - fp.x0, fp.w);
    s_off[sidx] = live ? make_int4(ly0 + lx0, ly0 + lx1, ly1 + lx0, ly1 + lx1) : make_int4(0, 0, 0, 0);
    s_wgt[sidx] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if (fsz <= kWaveFloats && nb <= 64 && spb <= 4) {
    // ---- small footprint (the common case): every WAVE stages and reduces its own quarter of
    // the block's channels in a private LDS region -- no block barriers after this one, so
    // the four waves (and the other blocks of the CU) drift apart and one wave's DMA latency
    // hides behind another's gathers.  A lane owns one bin: its <= 4 samples' tap offsets
    // and weights are read from the table ONCE into registers and reused for every channel.
    __syncthreads();   // table complete
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *wt = tile + wave * (kWaveFloats + kWavePad);
    const int per_wave = (c_end - c_block + 3) >> 2;
    const int cw0 = c_block + wave * per_wave, cw1 = min(c_end, cw0 + per_wave);
    int4 so[4];
    float4 sw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool on = lane < nb && q < spb;
      so[q] = on ? s_off[lane * spb + q] : make_int4(0, 0, 0, 0);
      sw[q] = on ? s_wgt[lane * spb + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int cpw = kWaveFloats / fsz;               // channels per pass (>= 1)
    const uint32_t plane32w = (uint32_t)plane;
    for (int cp = cw0; cp < cw1; cp += cpw) {
      const int nch = min(cpw, cw1 - cp);
      const float *src = feat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
      const int total = nch * fsz;
      if (X4) {
        const int q4 = fsz >> 2, rw4 = fp.w >> 2, total4 = total >> 2;
        const uint32_t m_q4 = q4 > 1 ? 0xFFFFFFFFu / (uint32_t)q4 + 1u : 0u;
        const uint32_t m_r4 = rw4 > 1 ? 0xFFFFFFFFu / (uint32_t)rw4 + 1u : 0u;
        for (int base = 0; base < total4; base += 64) {
          const uint32_t idx = (uint32_t)min(base + lane, total4 - 1);
          const uint32_t c = q4 > 1 ? __umulhi(idx, m_q4) : idx;
          const uint32_t r = idx - c * (uint32_t)q4;
          const uint32_t y = rw4 > 1 ? __umulhi(r, m_r4) : r;
          const uint32_t x = (r - y * (uint32_t)rw4) << 2;
          const uint32_t off = c * plane32w + y * (uint32_t)W + x;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(wt + base * 4), 16, 0, 0);
        }
      } else {
        const uint32_t m_fszw = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;
        const uint32_t m_ww = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
        for (int base = 0; base < total; base += 64) {
          const uint32_t idx = (uint32_t)min(base + lane, total - 1);
          const uint32_t c = fsz > 1 ? __umulhi(idx, m_fszw) : idx;
          const uint32_t r = idx - c * (uint32_t)fsz;
          const uint32_t y = fp.w > 1 ? __umulhi(r, m_ww) : r;
          const uint32_t x = r - y * (uint32_t)fp.w;
          const uint32_t off = c * plane32w + y * (uint32_t)W + x;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(wt + base), 4, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      for (int c = 0; c < nch; ++c) {
        const float *p = wt + c * fsz;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < spb) {
            float v = sw[q].x * p[so[q].x];
            v = v + sw[q].y * p[so[q].y];
            v = v + sw[q].z * p[so[q].z];
            v = v + sw[q].w * p[so[q].w];
            acc = acc + v;
          }
        }
        if (lane < nb) out[((k * C + cp + c) * (int64_t)nb) + lane] = acc / f.cnt;
      }
      // the next pass overwrites this region: its gathers are done (their values were consumed)
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const int cpp = kStageFloats / fsz;                // channels per pass (>= 1)
  const uint32_t m_nb = nb > 1 ? 0xFFFFFFFFu / (uint32_t)nb + 1u : 0u;  // it / nb for it < 2^16
  const uint32_t m_fsz = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;
  const uint32_t m_w = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
  const uint32_t plane32 = (uint32_t)plane;
  for (int cp = c_block; cp < c_end; cp += cpp) {
    const int nch = min(cpp, c_end - cp);
    // stage nch footprints with LDS-DMA (global_load_lds_dword): every lane supplies
    // its own source address, a wave-instruction lands 64 consecutive floats in the
    // tile, nothing passes through VGPRs and all loads are in flight together.
    // flat index -> (channel, y, x) by exact multiply-high division (idx < 2^14);
    // 32-bit element offsets from a uniform base.
    const float *src = feat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int total = nch * fsz;
      if (X4) {
        // 16 bytes per lane: one wave-instruction lands 1 KiB (64 chunks of 4 floats)
        const int q4 = fsz >> 2, rw4 = fp.w >> 2, total4 = total >> 2;   // chunks per plane / row / pass
        const uint32_t m_q4 = q4 > 1 ? 0xFFFFFFFFu / (uint32_t)q4 + 1u : 0u;
        const uint32_t m_r4 = rw4 > 1 ? 0xFFFFFFFFu / (uint32_t)rw4 + 1u : 0u;
        for (int base = wave * 64; base < total4; base += 256) {
          const uint32_t idx = (uint32_t)min(base + lane, total4 - 1);
          const uint32_t c = q4 > 1 ? __umulhi(idx, m_q4) : idx;
          const uint32_t r = idx - c * (uint32_t)q4;
          const uint32_t y = rw4 > 1 ? __umulhi(r, m_r4) : r;
          const uint32_t x = (r - y * (uint32_t)rw4) << 2;
          const uint32_t off = c * plane32 + y * (uint32_t)W + x;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(tile + base * 4), 16, 0, 0);
        }
      } else {
        for (int base = wave * 64; base < total; base += 256) {
          const uint32_t idx = (uint32_t)min(base + lane, total - 1);  // tail lanes re-load the last float into the pad
          const uint32_t c = fsz > 1 ? __umulhi(idx, m_fsz) : idx;
          const uint32_t r = idx - c * (uint32_t)fsz;
          const uint32_t y = fp.w > 1 ? __umulhi(r, m_w) : r;
          const uint32_t x = r - y * (uint32_t)fp.w;
          const uint32_t off = c * plane32 + y * (uint32_t)W + x;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(tile + base), 4, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // one output element per item: consecutive threads = consecutive bins of a channel
    // (coalesced stores); tap offsets / weights come from the block's sample table
    for (int it = threadIdx.x; it < nch * nb; it += 256) {
      const int c = nb > 1 ? (int)__umulhi((uint32_t)it, m_nb) : it, bin = it - c * nb;
      const float *p = tile + c * fsz;
      float acc = 0.0f;
      for (int q = 0; q < spb; ++q) {
        const int4 l = s_off[bin * spb + q];
        const float4 w = s_wgt[bin * spb + q];
        float v = w.x * p[l.x];
        v = v + w.y * p[l.y];
        v = v + w.z * p[l.z];
        v = v + w.w * p[l.w];
        acc = acc + v;
      }
      out[((k * C + cp + c) * (int64_t)nb) + bin] = acc / f.cnt;
    }
    __syncthreads();
  }
}

// ---- backward, direct atomics (any footprint size) -----------------------------------------
__device__ __forceinline__ void rroi_backward_direct(const float *__restrict__ gout, int C, int H, int W,
                                                     const RoiFrame &f, int64_t k, int ph, int pw,
                                                     int c_block, int groups_per_block,
                                                     float *__restrict__ gfeat) {
  const int nb = ph * pw;
  const int items = groups_per_block * nb;
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

// ---- backward, staged -----------------------------------------------------------------------
// Mirror image of the forward kernel: a wave accumulates the gradient of its channels in a
// private LDS copy of the RoI's footprint (ds_add_f32: the ~16 taps per output that neighbouring
// bins share meet in LDS, not in L2), then adds the footprint to the gradient map -- one
// global atomic per TOUCHED pixel and channel, in rows of consecutive addresses, instead of
// one scattered atomic per tap (3.1 -> see DESIGN.md for the measured time).  Float atomics
// commute only up to rounding: the tolerance of the op is 1e-4 * max|g| as before.
__global__ __launch_bounds__(256) void rroi_backward_kernel(
    const float *__restrict__ gout, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int groups_per_block,
    int K, int chunks, int roi_kind, float *__restrict__ gfeat) {
  __shared__ __attribute__((aligned(16))) float tile[4 * kWaveFloats];
  __shared__ int4 s_off[kMaxTabSamples];
  __shared__ float4 s_wgt[kMaxTabSamples];
  const uint32_t lin = blockIdx.x;          // same XCD-aware numbering as the forward kernel
  const uint32_t c8 = lin & 7u, rest = lin >> 3;
  const int64_t k = rest % (uint32_t)K;
  const int chunk = (int)((rest / (uint32_t)K) * 8u + c8);
  if (chunk >= chunks) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  if (f.b < 0 || f.b >= B) return;
  const int nb = ph * pw;
  const int c_block = chunk * groups_per_block * kChanPerThread;
  const int c_end = min(C, c_block + groups_per_block * kChanPerThread);
  const int64_t plane = (int64_t)H * W;
  const Footprint fp = roi_footprint(f, ph, pw, H, W);
  const int fsz = fp.w * fp.h;
  const int spb = f.gh * f.gw;
  const int nsamp = nb * spb;
  if (fsz > kWaveFloats || nb > 64 || spb > 4 || nsamp > kMaxTabSamples || (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_backward_direct(gout, C, H, W, f, k, ph, pw, c_block, groups_per_block, gfeat);
    return;
  }
  for (int sidx = threadIdx.x; sidx < nsamp; sidx += 256) {
    const int bin = sidx / spb, q = sidx - bin * spb;
    const int bi = bin / pw, bj = bin - bi * pw;
    const int iy = q / f.gw, ix = q - iy * f.gw;
    float x, y, w[4];
    int yl, xl, yh, xh;
    sample_xy(f, bi, bj, iy, ix, x, y);
    const bool live = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    const int ly0 = cl(yl - fp.y0, fp.h) * fp.w, ly1 = cl(yh - fp.y0, fp.h) * fp.w;
    const int lx0 = cl(xl - fp.x0, fp.w), lx1 = cl(xh - fp.x0, fp.w);
    s_off[sidx] = live ? make_int4(ly0 + lx0, ly0 + lx1, ly1 + lx0, ly1 + lx1) : make_int4(0, 0, 0, 0);
    s_wgt[sidx] = make_float4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();   // table complete; from here on the waves run independently
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *wt = tile + wave * kWaveFloats;
  const int per_wave = (c_end - c_block + 3) >> 2;
  const int cw0 = c_block + wave * per_wave, cw1 = min(c_end, cw0 + per_wave);
  int4 so[4];
  float4 sw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool on = lane < nb && q < spb;
    so[q] = on ? s_off[lane * spb + q] : make_int4(0, 0, 0, 0);
    sw[q] = on ? s_wgt[lane * spb + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int cpw = kWaveFloats / fsz;               // channels per pass (>= 1)
  const uint32_t m_fsz = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;
  const uint32_t m_w = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
  for (int cp = cw0; cp < cw1; cp += cpw) {
    const int nch = min(cpw, cw1 - cp);
    const int total = nch * fsz;
    for (int i = lane; i < total; i += 64) wt[i] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    for (int c = 0; c < nch; ++c) {
      float *p = wt + c * fsz;
      const float g = (lane < nb) ? gout[((k * C + cp + c) * (int64_t)nb) + lane] : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < spb && lane < nb) {
          // same per-tap arithmetic as the direct kernel: (g * w) / count
          atomicAdd(p + so[q].x, (g * sw[q].x) / f.cnt);
          atomicAdd(p + so[q].y, (g * sw[q].y) / f.cnt);
          atomicAdd(p + so[q].z, (g * sw[q].z) / f.cnt);
          atomicAdd(p + so[q].w, (g * sw[q].w) / f.cnt);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    float *dst = gfeat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    for (int i = lane; i < total; i += 64) {
      const float v = wt[i];
      if (v != 0.0f) {
        const uint32_t c = fsz > 1 ? __umulhi((uint32_t)i, m_fsz) : (uint32_t)i;
        const uint32_t r = (uint32_t)i - c * (uint32_t)fsz;
        const uint32_t y = fp.w > 1 ? __umulhi(r, m_w) : r;
        const uint32_t x = r - y * (uint32_t)fp.w;
        atomicAdd(dst + (int64_t)c * plane + (int64_t)y * W + x, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
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

static int rroi_forward_launch(const char *who, const float *feat_dev, int B, int C, int H, int W,
                               const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale,
                               int sample_num, float min_size, int roi_kind, float *out_dev, void *stream) {
  int rc = rroi_check(who, B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (K == 0 || C == 0) return OBB_OK;
  if (!feat_dev || !rois_dev || !out_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (H == 0 || W == 0 || B == 0) return set_err(OBB_EINVAL, "%s: empty feature map with %lld RoIs", who, (long long)K);
  // 16-32 channels per block (a big rotated RoI needs several staging passes per channel
  // group and the kernel ends with its slowest block), preferring a chunk count that is a
  // multiple of 8 = one chunk per XCD and round (see the kernel)
  const int groups = (C + kChanPerThread - 1) / kChanPerThread;
  int gpb = (groups % 32 == 0) ? 4 : ((groups % 16 == 0) ? 2 : 3);
  { const int64_t g = tuning(OBB_TUNE_RROI_GROUPS, gpb); if (g > 0 && g <= 64) gpb = (int)g; }
  const unsigned chunks = (unsigned)((groups + gpb - 1) / gpb);
  if (chunks > 65535u) return set_err(OBB_EINVAL, "%s: too many channels (%d)", who, C);
  // 16-byte LDS-DMA needs 16-byte aligned rows: W % 4 == 0 and an aligned base
  const bool x4 = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(feat_dev) & 15u) == 0);
  const unsigned long long nblk = (unsigned long long)((chunks + 7u) / 8u) * 8ull * (unsigned long long)K;
  if (nblk > 2147483647ull) return set_err(OBB_EINVAL, "%s: too many blocks (K=%lld, C=%d)", who, (long long)K, C);
  if (x4)
    hipLaunchKernelGGL(rroi_forward_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       feat_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, gpb, (int)K, (int)chunks, roi_kind, out_dev);
  else
    hipLaunchKernelGGL(rroi_forward_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       feat_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, gpb, (int)K, (int)chunks, roi_kind, out_dev);
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
  // same block geometry as the forward kernel (32 channels per block when C allows)
  const int groups = (C + kChanPerThread - 1) / kChanPerThread;
  int gpb = (groups % 32 == 0) ? 4 : ((groups % 16 == 0) ? 2 : 3);
  { const int64_t g = tuning(OBB_TUNE_RROI_GROUPS, gpb); if (g > 0 && g <= 64) gpb = (int)g; }
  const unsigned chunks = (unsigned)((groups + gpb - 1) / gpb);
  if (chunks > 65535u) return set_err(OBB_EINVAL, "%s: too many channels (%d)", who, C);
  const unsigned long long nblk = (unsigned long long)((chunks + 7u) / 8u) * 8ull * (unsigned long long)K;
  if (nblk > 2147483647ull) return set_err(OBB_EINVAL, "%s: too many blocks (K=%lld, C=%d)", who, (long long)K, C);
  hipLaunchKernelGGL(rroi_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     grad_out_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, gpb, (int)K, (int)chunks, roi_kind, grad_feat_dev);
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
