// obb_overlaps.hip -- pairwise rotated / polygon overlaps for gfx950.
//
// Replaces bbox_overlaps (mmdet/core/bbox/geometry.py:4-88) for (.,5) theta-OBBs
// and (.,8) polygons, and -- OBB_KIND_HBB -- for the (.,4) horizontal boxes themselves, in the
// reference's own arithmetic (hbb_pair_overlap, obb_geom.h).  Output is the dense (m,n) fp32 matrix, so the kernel is
// HBM-write bound when most pairs are disjoint; the layout is chosen for that:
//
//   wave = 256 consecutive columns (4 per lane -> one 16-byte store per lane,
//          1 KiB per wave-instruction), walking TI rows;
//   scan: a conservative 4-compare interval test per pair against the row's
//          interval broadcast from LDS; disjoint pairs store 0 immediately;
//   queue: surviving (row, col) pairs are compacted into a per-wave LDS queue
//          with ballot/popcount; whenever 64 are pending the whole wave clips
//          64 pairs at once (full lane utilisation for the ~1000-instruction
//          polygon clip instead of 1-2 live lanes per wave).
#include "obb_common.h"
#include "obb_geom.h"

namespace obb {

typedef float fvec4 __attribute__((ext_vector_type(4)));
constexpr int kWaves = 4;            // waves per block
constexpr int kColsPerLane = 4;
constexpr int kColsPerWave = 64 * kColsPerLane;      // 256
constexpr int kColsPerBlock = kWaves * kColsPerWave;  // 1024
constexpr int kMaxRows = 256;        // rows per block (TI upper bound)
constexpr int kQueue = 64 + kColsPerWave;  // carry (<= 63) + one row's worst case
constexpr uint32_t kGuard = 0x80008000u;
constexpr uint32_t kQMax = 32766u;

// conservative closed interval of a record along one axis (see DESIGN.md:
// margin 2^-21 * (|o| + e) dominates every rounding in the spec's prefilter)
__device__ __forceinline__ void interval(float o, float e, float &lo, float &hi) {
  float mg = (fabs_sel(o) + e) * 4.76837158203125e-7f;
  lo = (o - e) - mg;
  hi = (o + e) + mg;
}

// packed x|y 15-bit interval test with guard bits, 3 full-rate VALU per pair
// (same construction as the NMS scan kernel): 0 <=> the intervals meet on both axes
__device__ __forceinline__ uint32_t qdisjoint(uint32_t alo, uint32_t ahiG, uint32_t blo, uint32_t bhiG) {
  const uint32_t d1 = bhiG - alo, d2 = ahiG - blo;
  return (~d1 | ~d2) & kGuard;
}

template <int KIND>
__device__ __forceinline__ void clip_and_store(const float *__restrict__ a,
                                               const float *__restrict__ b, int64_t i,
                                               int64_t j, int64_t n, int mode,
                                               float *__restrict__ out) {
  constexpr int S = KIND == 0 ? 5 : (KIND == 1 ? 8 : 4);
  if (KIND == 2) {
    out[i * n + j] = hbb_pair_overlap(a + i * S, b + j * S, mode);
    return;
  }
  Quad A, B;
  load_quad<KIND>(a + i * S, A);
  load_quad<KIND>(b + j * S, B);
  out[i * n + j] = quad_overlap(A, B, mode);
}

// Block = 4 waves x 256 columns (4 per lane) x up to 256 rows.
//  1 every thread builds the float AABB interval of one row and of its 4 columns;
//    a block reduction gives the frame (bounding box of the block's rows + columns);
//    intervals are quantised to 15 bits in that frame -- floor of a monotone map is
//    monotone, so the integer test never loses a real overlap
//  2 per row: broadcast (v_readlane), 12 VALU for 4 columns + vote; every lane
//    stores 16 bytes of exact zeros (1 KiB per wave-instruction, the HBM-bound part)
//  3 rows with a candidate are handled out of line: ballot-compacted (row, col)
//    pairs go to a per-wave LDS queue; 64 at a time are clipped with all lanes busy
//    and overwrite their zeros (release fence + wave barrier order the two stores)
template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void overlaps_tile_kernel(
    const float *__restrict__ a, int64_t m, const float *__restrict__ b, int64_t n,
    int mode, int rows_per_block, float *__restrict__ out) {
  constexpr int S = KIND == 0 ? 5 : (KIND == 1 ? 8 : 4);
  __shared__ uint2 qrow[kMaxRows];                // packed (lo, hi|G) per row
  __shared__ uint32_t queue[kWaves][kQueue];
  __shared__ float red[kWaves][4];
  __shared__ float frame[4];                      // base x, base y, scale x, scale y
  __shared__ uint32_t fin[kWaves * 64];           // the waves' leftover candidates, clipped together
  __shared__ int fin_n;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  const int ti = (int)min((int64_t)rows_per_block, m - i0);
  const int64_t jw = (int64_t)blockIdx.x * kColsPerBlock + wave * kColsPerWave;  // wave's first col
  const int64_t j0 = jw + lane * kColsPerLane;

  // ---- 1a: float intervals (non-finite boxes become "isolated": lo > hi)
  const float big = 3.0e38f;
  float rxl = big, rxh = -big, ryl = big, ryh = -big;   // this thread's row
  bool rfin = false;
  // OBB_KIND_HBB: the reference defines a value for EVERY pair, also for boxes the prefilter
  // cannot speak for (non-positive +1 extent: negative or zero areas give -0 / NaN; non-finite
  // coordinates propagate) -- such "irregular" boxes meet everything, i.e. all their pairs take
  // the exact arithmetic
  bool runiv = false;
  if ((int)threadIdx.x < ti) {
    Quad A;
    load_quad<KIND>(a + (i0 + threadIdx.x) * S, A);
    interval(A.ox, A.ex, rxl, rxh);
    interval(A.oy, A.ey, ryl, ryh);
    rfin = (fabs_sel(rxl) < big) && (fabs_sel(rxh) < big) && (fabs_sel(ryl) < big) && (fabs_sel(ryh) < big);
    if (KIND == 2 && !(rfin && hbb_regular(a + (i0 + threadIdx.x) * S))) { runiv = true; rfin = false; }
  }
  float cxl[kColsPerLane], cxh[kColsPerLane], cyl[kColsPerLane], cyh[kColsPerLane];
  bool cfin[kColsPerLane], cuniv[kColsPerLane];
#pragma unroll
  for (int c = 0; c < kColsPerLane; ++c) {
    cfin[c] = false;
    cuniv[c] = false;
    cxl[c] = big; cxh[c] = -big; cyl[c] = big; cyh[c] = -big;
    if (j0 + c < n) {
      Quad B;
      load_quad<KIND>(b + (j0 + c) * S, B);
      interval(B.ox, B.ex, cxl[c], cxh[c]);
      interval(B.oy, B.ey, cyl[c], cyh[c]);
      cfin[c] = (fabs_sel(cxl[c]) < big) && (fabs_sel(cxh[c]) < big) && (fabs_sel(cyl[c]) < big) && (fabs_sel(cyh[c]) < big);
      if (KIND == 2 && !(cfin[c] && hbb_regular(b + (j0 + c) * S))) { cuniv[c] = true; cfin[c] = false; }
    }
  }
  // ---- 1b: frame = bounding box of all finite intervals of the block
  float mnx = big, mxx = -big, mny = big, mxy = -big;
  if (rfin) { mnx = rxl; mxx = rxh; mny = ryl; mxy = ryh; }
#pragma unroll
  for (int c = 0; c < kColsPerLane; ++c)
    if (cfin[c]) {
      mnx = min_sel(mnx, cxl[c]); mxx = max_sel(mxx, cxh[c]);
      mny = min_sel(mny, cyl[c]); mxy = max_sel(mxy, cyh[c]);
    }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min_sel(mnx, __shfl_xor(mnx, o)); mxx = max_sel(mxx, __shfl_xor(mxx, o));
    mny = min_sel(mny, __shfl_xor(mny, o)); mxy = max_sel(mxy, __shfl_xor(mxy, o));
  }
  if (lane == 0) { red[wave][0] = mnx; red[wave][1] = mxx; red[wave][2] = mny; red[wave][3] = mxy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float x0 = red[0][0], x1 = red[0][1], y0 = red[0][2], y1 = red[0][3];
    for (int k = 1; k < kWaves; ++k) {
      x0 = min_sel(x0, red[k][0]); x1 = max_sel(x1, red[k][1]);
      y0 = min_sel(y0, red[k][2]); y1 = max_sel(y1, red[k][3]);
    }
    const float rx = x1 - x0, ry = y1 - y0;
    frame[0] = x0; frame[1] = y0;
    frame[2] = rx > 0.0f ? (float)kQMax / rx : 0.0f;
    frame[3] = ry > 0.0f ? (float)kQMax / ry : 0.0f;
  }
  __syncthreads();
  const float bx = frame[0], by = frame[1], sx = frame[2], sy = frame[3];
  auto qz = [](float x, float base, float s) {
    float u = floorf((x - base) * s);
    u = u < 0.0f ? 0.0f : (u > (float)kQMax ? (float)kQMax : u);
    return (uint32_t)u;
  };
  // ---- 1c: quantise (isolated / absent boxes: lo = max, hi = 0 -> never a candidate)
  if ((int)threadIdx.x < kMaxRows) {
    uint2 o = make_uint2(0x7FFF7FFFu, kGuard);
    if (rfin) {
      o.x = qz(rxl, bx, sx) | (qz(ryl, by, sy) << 16);
      o.y = (qz(rxh, bx, sx) | (qz(ryh, by, sy) << 16)) | kGuard;
    }
    if (runiv) o = make_uint2(0u, (kQMax | (kQMax << 16)) | kGuard);   // meets every interval
    qrow[threadIdx.x] = o;
  }
  uint2 col[kColsPerLane];
#pragma unroll
  for (int c = 0; c < kColsPerLane; ++c) {
    col[c] = make_uint2(0x7FFF7FFFu, kGuard);
    if (cfin[c]) {
      col[c].x = qz(cxl[c], bx, sx) | (qz(cyl[c], by, sy) << 16);
      col[c].y = (qz(cxh[c], bx, sx) | (qz(cyh[c], by, sy) << 16)) | kGuard;
    }
    if (cuniv[c]) col[c] = make_uint2(0u, (kQMax | (kQMax << 16)) | kGuard);
  }
  __syncthreads();
  if (threadIdx.x == 0) fin_n = 0;   // (published by the barrier at the end of the scan)
  const bool wave_on = jw < n;        // a wave beyond the matrix only takes part in the barriers
  const int64_t jb = (int64_t)blockIdx.x * kColsPerBlock;

  uint32_t *q = queue[wave];
  int qn = 0;  // wave-uniform
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  auto drain = [&](bool flush) {
    while (qn >= 64 || (flush && qn > 0)) {
      const int take = qn < 64 ? qn : 64;
      qn -= take;
      // order this wave's zero stores before the clip's stores to the same words
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
#ifndef OBB_OV_NOCLIP
      if (lane < take) {
        const uint32_t e = q[qn + lane];
        clip_and_store<KIND>(a, b, i0 + (e >> 10), jb + (e & 1023u), n, mode, out);
      }
#endif
    }
  };

  for (int g = 0; wave_on && g < ti; g += 64) {          // 64 rows per VGPR pair
    const uint2 rv = qrow[g + lane];
    const int rows = min(64, ti - g);
    for (int ib = 0; ib < rows; ib += 8) {
      uint32_t hit = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ii = ib + u;                 // rows beyond ti are sentinels
        const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)rv.x, ii);
        const uint32_t rhi = (uint32_t)__builtin_amdgcn_readlane((int)rv.y, ii);
        const uint32_t t0 = qdisjoint(rlo, rhi, col[0].x, col[0].y);
        const uint32_t t1 = qdisjoint(rlo, rhi, col[1].x, col[1].y);
        const uint32_t t2 = qdisjoint(rlo, rhi, col[2].x, col[2].y);
        const uint32_t t3 = qdisjoint(rlo, rhi, col[3].x, col[3].y);
        const uint32_t mn = min(min(t0, t1), min(t2, t3));
        hit |= __any(mn == 0u) ? (1u << u) : 0u;
        // disjoint pairs are exact zeros; candidates are overwritten by the clip
#ifndef OBB_OV_NOSTORE
        if (g + ii < ti) {
          float *orow = out + (i0 + g + ii) * n + j0;
          if (VEC) {
            // write-once stream: non-temporal stores (+8 % on the 1 GB output)
            if (j0 < n) __builtin_nontemporal_store(fvec4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<fvec4 *>(orow));
          } else {
#pragma unroll
            for (int c = 0; c < kColsPerLane; ++c)
              if (j0 + c < n) orow[c] = 0.0f;
          }
        }
#endif
      }
      while (hit) {  // some lane has a candidate in row g+ib+u
        const int u = __builtin_ctz(hit);
        hit &= hit - 1u;
        const int ii = ib + u;
        const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)rv.x, ii);
        const uint32_t rhi = (uint32_t)__builtin_amdgcn_readlane((int)rv.y, ii);
#pragma unroll
        for (int c = 0; c < kColsPerLane; ++c) {
          const bool cand = qdisjoint(rlo, rhi, col[c].x, col[c].y) == 0u;
          const uint64_t mk = __ballot(cand);
          if (mk) {
            if (cand) q[qn + __popcll(mk & lt_mask)] = ((uint32_t)(g + ii) << 10) | (uint32_t)(wave * kColsPerWave + lane * kColsPerLane + c);
            qn += __popcll(mk);
          }
        }
        drain(false);
      }
    }
  }
  // the < 64 candidates every wave has left are pooled and clipped 64 at a time by the whole
  // block (four partial passes of ~1100 VALU each become one or two full ones)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's zero stores are out
  __syncthreads();
  int base = 0;
  if (lane == 0 && qn > 0) base = atomicAdd(&fin_n, qn);
  base = __shfl(base, 0);
  if (lane < qn) fin[base + lane] = q[lane];
  __syncthreads();
  const int total = fin_n;
#ifndef OBB_OV_NOCLIP
  for (int start = wave * 64; start < total; start += kWaves * 64) {
    if (start + lane < total) {
      const uint32_t e = fin[start + lane];
      clip_and_store<KIND>(a, b, i0 + (e >> 10), jb + (e & 1023u), n, mode, out);
    }
  }
#endif
}

// aligned (pair i with pair i): one lane per pair
template <int KIND>
__global__ __launch_bounds__(256) void overlaps_aligned_kernel(
    const float *__restrict__ a, const float *__restrict__ b, int64_t n, int mode,
    float *__restrict__ out) {
  constexpr int S = KIND == 0 ? 5 : (KIND == 1 ? 8 : 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (KIND == 2) {
      out[i] = hbb_pair_overlap(a + i * S, b + i * S, mode);
      continue;
    }
    Quad A, B;
    load_quad<KIND>(a + i * S, A);
    load_quad<KIND>(b + i * S, B);
    out[i] = quad_overlap(A, B, mode);
  }
}

__global__ __launch_bounds__(256) void rbox_to_poly_kernel(const float *__restrict__ rb,
                                                           int64_t n, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    Quad q;
    const float *r = rb + i * 5;
    quad_from_rbox(r[0], r[1], r[2], r[3], r[4], q);
    // cv2.boxPoints order: c-u+v, c-u-v, c+u-v, c+u+v  == q.r[3], r[0], r[1], r[2]
    float *o = out + i * 8;
    o[0] = q.ox + q.rx[3]; o[1] = q.oy + q.ry[3];
    o[2] = q.ox + q.rx[0]; o[3] = q.oy + q.ry[0];
    o[4] = q.ox + q.rx[1]; o[5] = q.oy + q.ry[1];
    o[6] = q.ox + q.rx[2]; o[7] = q.oy + q.ry[2];
  }
}

// ---- float64 horizontal boxes ----------------------------------------------------------------------------------
// bbox_overlaps is dtype-generic in the reference (torch expressions, mmdet/core/bbox/geometry.py:47-88): double boxes
// give double overlaps.  Same expressions as hbb_pair_overlap (geometry.py:61-86 in its tensor-op order, NaNs travelling
// as torch.max / min / clamp move them), evaluated in double; one thread per output, consecutive threads = consecutive
// columns (8-byte coalesced stores: the op is write bound).
__device__ __forceinline__ double nan_max_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a < b ? b : a)); }
__device__ __forceinline__ double nan_min_d(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double hbb_pair_overlap_d(const double *__restrict__ a, const double *__restrict__ b, int mode) {
  const double ltx = nan_max_d(a[0], b[0]), lty = nan_max_d(a[1], b[1]);
  const double rbx = nan_min_d(a[2], b[2]), rby = nan_min_d(a[3], b[3]);
  const double w = nan_max_d((rbx - ltx) + 1.0, 0.0), h = nan_max_d((rby - lty) + 1.0, 0.0);
  const double overlap = w * h;
  const double area1 = ((a[2] - a[0]) + 1.0) * ((a[3] - a[1]) + 1.0);
  if (mode != 0) return overlap / area1;
  const double area2 = ((b[2] - b[0]) + 1.0) * ((b[3] - b[1]) + 1.0);
  return overlap / ((area1 + area2) - overlap);
}
__global__ __launch_bounds__(256) void hbb_overlaps_f64_kernel(const double *__restrict__ a, int64_t m, const double *__restrict__ b,
                                                               int64_t n, int mode, int aligned, double *__restrict__ out) {
  const int64_t total = aligned ? n : m * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = aligned ? t : t / n, j = aligned ? t : t - i * n;
    out[t] = hbb_pair_overlap_d(a + i * 4, b + j * 4, mode);
  }
}

}  // namespace obb

extern "C" int obb_hbb_overlaps_f64(const double *a_dev, int64_t m, const double *b_dev, int64_t n, int mode, int aligned,
                                    double *out_dev, void *stream) {
  using namespace obb;
  if (m < 0 || n < 0) return set_err(OBB_EINVAL, "obb_hbb_overlaps_f64: negative size");
  if (mode != OBB_MODE_IOU && mode != OBB_MODE_IOF) return set_err(OBB_EINVAL, "obb_hbb_overlaps_f64: bad mode %d", mode);
  if (aligned && m != n) return set_err(OBB_EINVAL, "obb_hbb_overlaps_f64: aligned needs m == n");
  if (m == 0 || n == 0) return OBB_OK;
  if (!a_dev || !b_dev || !out_dev) return set_err(OBB_EINVAL, "obb_hbb_overlaps_f64: null pointer");
  const int64_t total = aligned ? n : m * n;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(hbb_overlaps_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a_dev, m, b_dev, n, mode, aligned, out_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_overlaps_f32(const float *a_dev, int64_t m, const float *b_dev, int64_t n,
                                int kind, int mode, int aligned, float *out_dev, void *stream) {
  using namespace obb;
  if (m < 0 || n < 0) return set_err(OBB_EINVAL, "obb_overlaps_f32: negative size (m=%lld n=%lld)", (long long)m, (long long)n);
  if (kind != OBB_KIND_RBOX && kind != OBB_KIND_POLY && kind != OBB_KIND_HBB) return set_err(OBB_EINVAL, "obb_overlaps_f32: bad kind %d", kind);
  if (mode != OBB_MODE_IOU && mode != OBB_MODE_IOF) return set_err(OBB_EINVAL, "obb_overlaps_f32: bad mode %d", mode);
  if (aligned && m != n) return set_err(OBB_EINVAL, "obb_overlaps_f32: aligned needs m == n (%lld vs %lld)", (long long)m, (long long)n);
  if (m == 0 || n == 0) return OBB_OK;
  if (!a_dev || !b_dev || !out_dev) return set_err(OBB_EINVAL, "obb_overlaps_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (aligned) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (kind == OBB_KIND_RBOX)
      hipLaunchKernelGGL(overlaps_aligned_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, a_dev, b_dev, n, mode, out_dev);
    else if (kind == OBB_KIND_POLY)
      hipLaunchKernelGGL(overlaps_aligned_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a_dev, b_dev, n, mode, out_dev);
    else
      hipLaunchKernelGGL(overlaps_aligned_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, a_dev, b_dev, n, mode, out_dev);
    OBB_LAUNCH_CHECK();
    return OBB_OK;
  }
  const int64_t col_tiles = (n + kColsPerBlock - 1) / kColsPerBlock;
  // rows per block: measured (tools/exp_iou3.sh) -- 40..56 rows stream the output ~20 %
  // faster than 256 (and than 64: power-of-two row counts collide at a 64 KB pitch);
  // fewer rows while that leaves fewer than ~512 blocks
  int rows = 8;
  for (int r : {48, 32, 24, 16, 12, 8}) {
    rows = r;
    if (col_tiles * ((m + r - 1) / r) >= 512) break;
  }
  { const int64_t r = tuning(OBB_TUNE_OVERLAPS_ROWS, rows); if (r >= 8 && r <= kMaxRows) rows = (int)r; }
  const int64_t row_tiles2 = (m + rows - 1) / rows;
  if (row_tiles2 > 65535 || col_tiles > 2147483647LL)
    return set_err(OBB_EINVAL, "obb_overlaps_f32: problem too large for one launch (m=%lld n=%lld)", (long long)m, (long long)n);
  dim3 grid((unsigned)col_tiles, (unsigned)row_tiles2);
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0);
#define OBB_LAUNCH_TILE(K, V) \
  hipLaunchKernelGGL((overlaps_tile_kernel<K, V>), grid, dim3(256), 0, st, a_dev, m, b_dev, n, mode, rows, out_dev)
  if (kind == OBB_KIND_RBOX) { if (vec) OBB_LAUNCH_TILE(0, true); else OBB_LAUNCH_TILE(0, false); }
  else if (kind == OBB_KIND_POLY) { if (vec) OBB_LAUNCH_TILE(1, true); else OBB_LAUNCH_TILE(1, false); }
  else { if (vec) OBB_LAUNCH_TILE(2, true); else OBB_LAUNCH_TILE(2, false); }
#undef OBB_LAUNCH_TILE
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_rbox_to_poly_f32(const float *rbox_dev, int64_t n, float *poly_dev, void *stream) {
  using namespace obb;
  if (n < 0) return set_err(OBB_EINVAL, "obb_rbox_to_poly_f32: negative n");
  if (n == 0) return OBB_OK;
  if (!rbox_dev || !poly_dev) return set_err(OBB_EINVAL, "obb_rbox_to_poly_f32: null pointer");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rbox_to_poly_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rbox_dev, n, poly_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
