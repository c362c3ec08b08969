// obb_overlaps.hip -- pairwise rotated / polygon overlaps for gfx950.
//
// Replaces bbox_overlaps (mmdet/core/bbox/geometry.py:4-88) for (.,5) theta-OBBs
// and (.,8) polygons.  Output is the dense (m,n) fp32 matrix, so the kernel is
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
#include "obb_geom.cuh"

namespace obb {

constexpr int kWaves = 4;            // waves per block
constexpr int kColsPerLane = 4;
constexpr int kColsPerWave = 64 * kColsPerLane;      // 256
constexpr int kColsPerBlock = kWaves * kColsPerWave;  // 1024
constexpr int kMaxRows = 256;        // rows per block (TI upper bound)
constexpr int kQueue = 64 + kColsPerWave;  // worst case pending + one row's pushes

// conservative closed interval of a record along one axis (see DESIGN.md:
// margin 2^-21 * (|o| + e) dominates every rounding in the spec's prefilter)
__device__ __forceinline__ void interval(float o, float e, float &lo, float &hi) {
  float mg = (fabs_sel(o) + e) * 4.76837158203125e-7f;
  lo = (o - e) - mg;
  hi = (o + e) + mg;
}

template <int KIND>
__device__ __forceinline__ void clip_and_store(const float *__restrict__ a,
                                               const float *__restrict__ b, int64_t i,
                                               int64_t j, int64_t n, int mode,
                                               float *__restrict__ out) {
  constexpr int S = KIND ? 8 : 5;
  Quad A, B;
  load_quad<KIND>(a + i * S, A);
  load_quad<KIND>(b + j * S, B);
  out[i * n + j] = quad_overlap(A, B, mode);
}

template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void overlaps_tile_kernel(
    const float *__restrict__ a, int64_t m, const float *__restrict__ b, int64_t n,
    int mode, int rows_per_block, float *__restrict__ out) {
  constexpr int S = KIND ? 8 : 5;
  __shared__ float4 arow[kMaxRows];              // xlo, xhi, ylo, yhi per row
  __shared__ uint32_t queue[kWaves][kQueue];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  const int ti = (int)min((int64_t)rows_per_block, m - i0);
  const int64_t jw = (int64_t)blockIdx.x * kColsPerBlock + wave * kColsPerWave;  // wave's first col
  const int64_t j0 = jw + lane * kColsPerLane;

  // row intervals -> LDS
  if (threadIdx.x < ti) {
    Quad A;
    load_quad<KIND>(a + (i0 + threadIdx.x) * S, A);
    float4 r;
    interval(A.ox, A.ex, r.x, r.y);
    interval(A.oy, A.ey, r.z, r.w);
    arow[threadIdx.x] = r;
  }
  // this lane's 4 column intervals -> registers
  float bxlo[kColsPerLane], bxhi[kColsPerLane], bylo[kColsPerLane], byhi[kColsPerLane];
#pragma unroll
  for (int c = 0; c < kColsPerLane; ++c) {
    if (j0 + c < n) {
      Quad B;
      load_quad<KIND>(b + (j0 + c) * S, B);
      interval(B.ox, B.ex, bxlo[c], bxhi[c]);
      interval(B.oy, B.ey, bylo[c], byhi[c]);
    } else {  // beyond the last column: an interval no row can meet
      bxlo[c] = 3.0e38f; bxhi[c] = -3.0e38f; bylo[c] = 3.0e38f; byhi[c] = -3.0e38f;
    }
  }
  __syncthreads();
  if (jw >= n) return;  // whole wave beyond the matrix (after the barrier)

  uint32_t *q = queue[wave];
  int qn = 0;  // wave-uniform
  const uint64_t lt_mask = (1ull << lane) - 1ull;

  for (int ii = 0; ii < ti; ++ii) {
    const float4 r = arow[ii];
    bool cand[kColsPerLane];
#pragma unroll
    for (int c = 0; c < kColsPerLane; ++c)
      cand[c] = (r.x <= bxhi[c]) && (bxlo[c] <= r.y) && (r.z <= byhi[c]) && (bylo[c] <= r.w);

    // disjoint pairs are exact zeros; candidates are overwritten by the clip
    float *orow = out + (i0 + ii) * n + j0;
    if (VEC) {
      if (j0 < n) *reinterpret_cast<float4 *>(orow) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int c = 0; c < kColsPerLane; ++c)
        if (j0 + c < n) orow[c] = 0.0f;
    }

#pragma unroll
    for (int c = 0; c < kColsPerLane; ++c) {
      const uint64_t mk = __ballot(cand[c]);
      if (mk) {
        if (cand[c]) q[qn + __popcll(mk & lt_mask)] = ((uint32_t)ii << 10) | (uint32_t)(lane * kColsPerLane + c);
        qn += __popcll(mk);
      }
    }
    while (qn >= 64) {
      qn -= 64;
      // order this wave's zero stores before the clip's stores to the same words
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const uint32_t e = q[qn + lane];
      clip_and_store<KIND>(a, b, i0 + (e >> 10), jw + (e & 1023u), n, mode, out);
    }
  }
  if (qn > 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < qn) {
      const uint32_t e = q[lane];
      clip_and_store<KIND>(a, b, i0 + (e >> 10), jw + (e & 1023u), n, mode, out);
    }
  }
}

// aligned (pair i with pair i): one lane per pair
template <int KIND>
__global__ __launch_bounds__(256) void overlaps_aligned_kernel(
    const float *__restrict__ a, const float *__restrict__ b, int64_t n, int mode,
    float *__restrict__ out) {
  constexpr int S = KIND ? 8 : 5;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
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

}  // namespace obb

extern "C" int obb_overlaps_f32(const float *a_dev, int64_t m, const float *b_dev, int64_t n,
                                int kind, int mode, int aligned, float *out_dev, void *stream) {
  using namespace obb;
  if (m < 0 || n < 0) return set_err(OBB_EINVAL, "obb_overlaps_f32: negative size (m=%lld n=%lld)", (long long)m, (long long)n);
  if (kind != OBB_KIND_RBOX && kind != OBB_KIND_POLY) return set_err(OBB_EINVAL, "obb_overlaps_f32: bad kind %d", kind);
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
    else
      hipLaunchKernelGGL(overlaps_aligned_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a_dev, b_dev, n, mode, out_dev);
    OBB_LAUNCH_CHECK();
    return OBB_OK;
  }
  const int64_t col_tiles = (n + kColsPerBlock - 1) / kColsPerBlock;
  // rows per block: as many as possible while keeping >= ~1024 blocks in flight
  int rows = kMaxRows;
  while (rows > 8 && col_tiles * ((m + rows - 1) / rows) < 1024) rows >>= 1;
  const int64_t row_tiles2 = (m + rows - 1) / rows;
  if (row_tiles2 > 65535 || col_tiles > 2147483647LL)
    return set_err(OBB_EINVAL, "obb_overlaps_f32: problem too large for one launch (m=%lld n=%lld)", (long long)m, (long long)n);
  dim3 grid((unsigned)col_tiles, (unsigned)row_tiles2);
  const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0);
#define OBB_LAUNCH_TILE(K, V) \
  hipLaunchKernelGGL((overlaps_tile_kernel<K, V>), grid, dim3(256), 0, st, a_dev, m, b_dev, n, mode, rows, out_dev)
  if (kind == OBB_KIND_RBOX) { if (vec) OBB_LAUNCH_TILE(0, true); else OBB_LAUNCH_TILE(0, false); }
  else { if (vec) OBB_LAUNCH_TILE(1, true); else OBB_LAUNCH_TILE(1, false); }
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
