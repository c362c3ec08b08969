// obb_nms.hip -- greedy rotated / polygon NMS on gfx950, entirely on the GPU.
//
// Replaces nms_cuda (mmdet/ops/nms/src/nms_kernel.cu:71-139: sort, 64x64
// bitmask kernel, 1.25 GB mask D2H copy at n = 100k, serial host reduce) with
// the semantics of the CPU reference (mmdet/ops/nms/src/nms_cpu.cpp:21-59).
//
// Pipeline (one stream, no host round trip until the final count read-back):
//   1 key      score -> order-preserving uint32 key, value = original index
//   2 sort     stable LSD radix sort (rocPRIM) -> order[]
//   3 gather   boxes in sorted order -> 64-byte quad records + conservative
//              AABB intervals (float4), both coalesced for the pair kernel
//   4 pair     upper-triangular tiles; 4-compare interval scan per pair, LDS
//              ballot-compacted candidate queue, full-wave polygon clip; pairs
//              with overlap >= thr are appended to a sparse EDGE list (i -> j,
//              i before j in score order) and counted per target (cnt[j]).
//              The n x n/64 bit matrix of the reference is never materialised.
//   5 resolve  the greedy scan is the unique fixpoint of
//                 kept(j)  <=>  no kept i with an edge i -> j
//              evaluated by rounds over the live edges only: an edge whose
//              source is kept suppresses its target; an edge whose source is
//              suppressed decrements cnt[target], which becomes kept at zero.
//   6 compact  keep flags scattered to original index order, block scan ->
//              ascending int64 indices (nms_cpu.cpp:59).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "obb_common.h"
#include "obb_geom.cuh"

namespace obb {

constexpr int kNmsWaves = 4;
constexpr int kNmsColsPerLane = 4;
constexpr int kNmsColsPerWave = 64 * kNmsColsPerLane;           // 256
constexpr int kNmsColsPerBlock = kNmsWaves * kNmsColsPerWave;   // 1024
constexpr int kNmsMaxRows = 256;
constexpr int kNmsQueue = 64 + kNmsColsPerWave;

constexpr uint8_t kUndecided = 0, kKept = 1, kSuppressed = 2;

struct alignas(16) QuadRec {  // 64 bytes: four float4 loads
  float ox, oy, ex, ey;
  float rx[4];
  float ry[4];
  float area, pad0, pad1, pad2;
};

struct NmsCounters {
  unsigned long long edges;   // overlap edges found (may exceed capacity)
  long long kept;             // K
  long long rounds;           // resolve rounds
  long long pad;
};

__device__ __forceinline__ void nms_interval(float o, float e, float &lo, float &hi) {
  float mg = (fabs_sel(o) + e) * 4.76837158203125e-7f;
  lo = (o - e) - mg;
  hi = (o + e) + mg;
}

// ---- 1: keys ---------------------------------------------------------------
__global__ __launch_bounds__(256) void nms_key_kernel(const float *__restrict__ dets, int64_t n,
                                                      int stride, uint32_t *__restrict__ keys,
                                                      uint32_t *__restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = score_key(dets[i * stride + stride - 1]);
    vals[i] = (uint32_t)i;
  }
}

// ---- 3: gather into sorted order --------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void nms_gather_kernel(const float *__restrict__ dets, int64_t n,
                                                         const uint32_t *__restrict__ order,
                                                         QuadRec *__restrict__ rec,
                                                         float4 *__restrict__ iv) {
  constexpr int S = KIND ? 9 : 6;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
       p += (int64_t)gridDim.x * blockDim.x) {
    Quad q;
    load_quad<KIND>(dets + (int64_t)order[p] * S, q);
    QuadRec r;
    r.ox = q.ox; r.oy = q.oy; r.ex = q.ex; r.ey = q.ey;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.rx[k] = q.rx[k]; r.ry[k] = q.ry[k]; }
    r.area = q.area; r.pad0 = 0.f; r.pad1 = 0.f; r.pad2 = 0.f;
    rec[p] = r;
    float4 v;
    nms_interval(q.ox, q.ex, v.x, v.y);
    nms_interval(q.oy, q.ey, v.z, v.w);
    iv[p] = v;
  }
}

__device__ __forceinline__ void load_rec(const QuadRec *__restrict__ rec, uint32_t p, Quad &q) {
  const float4 *r = reinterpret_cast<const float4 *>(rec + p);
  float4 a = r[0], b = r[1], c = r[2], d = r[3];
  q.ox = a.x; q.oy = a.y; q.ex = a.z; q.ey = a.w;
  q.rx[0] = b.x; q.rx[1] = b.y; q.rx[2] = b.z; q.rx[3] = b.w;
  q.ry[0] = c.x; q.ry[1] = c.y; q.ry[2] = c.z; q.ry[3] = c.w;
  q.area = d.x;
}

// clip one candidate pair per lane; append overlap edges (wave-aggregated)
__device__ __forceinline__ void nms_clip_emit(const QuadRec *__restrict__ rec, bool active,
                                              uint32_t i, uint32_t j, float thr,
                                              uint2 *__restrict__ edges, unsigned long long cap,
                                              NmsCounters *__restrict__ ctr, int *__restrict__ cnt,
                                              int lane) {
  bool hit = false;
  if (active) {
    Quad A, B;
    load_rec(rec, i, A);
    load_rec(rec, j, B);
    float ov = quad_overlap(A, B, 0);
    hit = ov >= thr;
  }
  const uint64_t mk = __ballot(hit);
  if (mk) {
    const int leader = __ffsll((long long)mk) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&ctr->edges, (unsigned long long)__popcll(mk));
    base = __shfl(base, leader);
    if (hit) {
      const unsigned long long pos = base + __popcll(mk & ((1ull << lane) - 1ull));
      if (pos < cap) edges[pos] = make_uint2(i, j);
      atomicAdd(&cnt[j], 1);
    }
  }
}

// ---- 4: pair kernel -----------------------------------------------------------
// Block = 4 waves x 256 columns; rows_per_block rows.  Blocks enumerate only
// the tiles that intersect the strict upper triangle (j > i).
template <bool DIAG>
__device__ __forceinline__ void nms_scan_rows(const float4 *arow, int ti, uint32_t i0, uint32_t jw,
                                              uint32_t j0, const float (&bxlo)[4],
                                              const float (&bxhi)[4], const float (&bylo)[4],
                                              const float (&byhi)[4], uint32_t *q,
                                              const QuadRec *__restrict__ rec, float thr,
                                              uint2 *__restrict__ edges, unsigned long long cap,
                                              NmsCounters *__restrict__ ctr, int *__restrict__ cnt,
                                              int lane) {
  int qn = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (int ii = 0; ii < ti; ++ii) {
    const float4 r = arow[ii];
#pragma unroll
    for (int c = 0; c < kNmsColsPerLane; ++c) {
      bool cand = (r.x <= bxhi[c]) && (bxlo[c] <= r.y) && (r.z <= byhi[c]) && (bylo[c] <= r.w);
      if (DIAG) cand = cand && (j0 + c > i0 + ii);
      const uint64_t mk = __ballot(cand);
      if (mk) {
        if (cand) q[qn + __popcll(mk & lt_mask)] = ((uint32_t)ii << 10) | (uint32_t)(lane * kNmsColsPerLane + c);
        qn += __popcll(mk);
      }
    }
    while (qn >= 64) {
      qn -= 64;
      __builtin_amdgcn_wave_barrier();
      const uint32_t e = q[qn + lane];
      nms_clip_emit(rec, true, i0 + (e >> 10), jw + (e & 1023u), thr, edges, cap, ctr, cnt, lane);
    }
  }
  if (qn > 0) {
    __builtin_amdgcn_wave_barrier();
    const bool act = lane < qn;
    const uint32_t e = act ? q[lane] : 0u;
    nms_clip_emit(rec, act, i0 + (e >> 10), jw + (e & 1023u), thr, edges, cap, ctr, cnt, lane);
  }
}

__global__ __launch_bounds__(256) void nms_pair_kernel(const float4 *__restrict__ iv,
                                                       const QuadRec *__restrict__ rec, uint32_t n,
                                                       float thr, int rows_per_block, int nj,
                                                       uint2 *__restrict__ edges,
                                                       unsigned long long cap,
                                                       NmsCounters *__restrict__ ctr,
                                                       int *__restrict__ cnt) {
  __shared__ float4 arow[kNmsMaxRows];
  __shared__ uint32_t queue[kNmsWaves][kNmsQueue];

  // block id -> (row tile bi, col tile bj).  G = 1024/rows row tiles share the
  // same first column tile g; blocks before group g: G*(g*nj - g(g-1)/2).
  const int G = kNmsColsPerBlock / rows_per_block;
  const long long id = blockIdx.x;
  const double twoNj1 = 2.0 * nj + 1.0;
  long long g = (long long)((twoNj1 - sqrt(twoNj1 * twoNj1 - 8.0 * (double)id / G)) * 0.5);
  if (g < 0) g = 0;
  if (g > nj - 1) g = nj - 1;
  auto before = [&](long long gg) { return (long long)G * (gg * nj - gg * (gg - 1) / 2); };
  while (g > 0 && before(g) > id) --g;
  while (g + 1 < nj && before(g + 1) <= id) ++g;
  const long long rem = id - before(g);
  const int width = nj - (int)g;            // col tiles per row tile in this group
  const uint32_t bi = (uint32_t)(g * G + rem / width);
  const uint32_t bj = (uint32_t)(g + rem % width);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t i0 = bi * (uint32_t)rows_per_block;
  if (i0 >= n) return;  // (partial last group)
  const int ti = (int)min((uint32_t)rows_per_block, n - i0);
  const uint32_t jw = bj * kNmsColsPerBlock + wave * kNmsColsPerWave;
  const uint32_t j0 = jw + lane * kNmsColsPerLane;

  for (int t = threadIdx.x; t < ti; t += 256) arow[t] = iv[i0 + t];
  float bxlo[4], bxhi[4], bylo[4], byhi[4];
#pragma unroll
  for (int c = 0; c < kNmsColsPerLane; ++c) {
    if (j0 + c < n) {
      const float4 v = iv[j0 + c];
      bxlo[c] = v.x; bxhi[c] = v.y; bylo[c] = v.z; byhi[c] = v.w;
    } else {
      bxlo[c] = 3.0e38f; bxhi[c] = -3.0e38f; bylo[c] = 3.0e38f; byhi[c] = -3.0e38f;
    }
  }
  __syncthreads();
  if (jw >= n) return;
  if (jw + kNmsColsPerWave - 1 <= i0) return;  // every column <= every row: no pair j > i

  if (jw <= i0 + (uint32_t)ti - 1)
    nms_scan_rows<true>(arow, ti, i0, jw, j0, bxlo, bxhi, bylo, byhi, queue[wave], rec, thr, edges, cap, ctr, cnt, lane);
  else
    nms_scan_rows<false>(arow, ti, i0, jw, j0, bxlo, bxhi, bylo, byhi, queue[wave], rec, thr, edges, cap, ctr, cnt, lane);
}

// ---- 5: resolve ---------------------------------------------------------------
// agent-scope loads (L2): cnt is updated by atomics, which bypass this CU's L1
__device__ __forceinline__ uint8_t ld_state(uint8_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_cnt(int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup walks the live edge list until it is empty, compacting it in
// place each round (reads of a 1024-edge chunk complete before its writes, and
// the write cursor never passes the chunk being read).
__global__ __launch_bounds__(1024) void nms_resolve_kernel(uint2 *__restrict__ edges,
                                                           unsigned long long cap,
                                                           uint8_t *__restrict__ state,
                                                           int *__restrict__ cnt,
                                                           NmsCounters *__restrict__ ctr) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long found = ctr->edges;
  // an overflowed edge list is incomplete (cnt[] counts edges that were not
  // stored): nothing to resolve, the host reports OBB_EWORKSPACE
  long long live = found <= cap ? (long long)found : 0;
  long long rounds = 0;
  int stalled = 0;
  while (live > 0) {
    long long w = 0;
    for (long long base = 0; base < live; base += 1024) {
      const long long e = base + tid;
      bool keep = false;
      uint2 ed = make_uint2(0u, 0u);
      if (e < live) {
        ed = edges[e];
        if (ld_state(state + ed.y) == kUndecided) {
          uint8_t si = ld_state(state + ed.x);
          if (si == kUndecided && ld_cnt(cnt + ed.x) == 0) si = kKept;
          if (si == kKept) {
            state[ed.y] = kSuppressed;
          } else if (si == kSuppressed) {
            if (atomicSub(&cnt[ed.y], 1) == 1) state[ed.y] = kKept;
          } else {
            keep = true;
          }
        }
      }
      const uint64_t mk = __ballot(keep);
      if (lane == 0) wsum[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int v = wsum[k];
        off += (k < wave) ? v : 0;
        tot += v;
      }
      if (keep) edges[w + off + __popcll(mk & ((1ull << lane) - 1ull))] = ed;
      w += tot;
      __syncthreads();
    }
    // every round retires at least the live edge with the smallest source (its
    // source is decided); two rounds without progress can only mean corrupted
    // input state -- stop instead of spinning
    stalled = (w == live) ? stalled + 1 : 0;
    live = w;
    ++rounds;
    __threadfence();
    __syncthreads();
    if (stalled >= 2) break;
  }
  if (tid == 0) ctr->rounds = rounds;
}

// ---- 6: compaction --------------------------------------------------------------
__global__ __launch_bounds__(256) void nms_flag_kernel(const uint32_t *__restrict__ order,
                                                       const uint8_t *__restrict__ state,
                                                       const int *__restrict__ cnt, int64_t n,
                                                       uint8_t *__restrict__ keepflag) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n;
       p += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t s = state[p];
    keepflag[order[p]] = (s == kKept) || (s == kUndecided && cnt[p] == 0);
  }
}

__global__ __launch_bounds__(1024) void nms_count_kernel(const uint8_t *__restrict__ keepflag,
                                                         int64_t n, int *__restrict__ blockcnt) {
  __shared__ int wsum[16];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const bool f = (i < n) && keepflag[i];
  const uint64_t mk = __ballot(f);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(mk);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += wsum[k];
    blockcnt[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void nms_write_kernel(const uint8_t *__restrict__ keepflag,
                                                         int64_t n, const int *__restrict__ blockcnt,
                                                         int64_t *__restrict__ keep,
                                                         NmsCounters *__restrict__ ctr) {
  __shared__ int wsum[16];
  __shared__ long long sbase;
  __shared__ long long red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // base = sum of the counts of all previous blocks
  long long part = 0;
  for (int b = tid; b < (int)blockIdx.x; b += 1024) part += blockcnt[b];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
  if (lane == 0) red[wave] = part;
  const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
  const bool f = (i < n) && keepflag[i];
  const uint64_t mk = __ballot(f);
  if (lane == 0) wsum[wave] = __popcll(mk);
  __syncthreads();
  if (tid == 0) {
    long long t = 0;
    for (int k = 0; k < 16; ++k) t += red[k];
    sbase = t;
  }
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int v = wsum[k];
    off += (k < wave) ? v : 0;
    tot += v;
  }
  if (f) keep[sbase + off + __popcll(mk & ((1ull << lane) - 1ull))] = i;
  if (blockIdx.x == gridDim.x - 1 && tid == 0) ctr->kept = sbase + tot;
}

// ---- optional per-phase timing (HIP events on the caller's stream) ----------------
constexpr int kPhases = OBB_NMS_PHASES;
struct NmsProfile {
  bool on = false;
  bool have = false;
  hipEvent_t ev[kPhases + 1] = {};
  bool created = false;
  float ms[kPhases] = {};
};
static thread_local NmsProfile g_prof;

// ---- workspace layout ------------------------------------------------------------
struct NmsLayout {
  size_t keys_in, keys_out, vals_in, order, iv, rec, zero_begin, state, cnt, ctr, zero_end,
      keepflag, blockcnt, sort_temp, edges, total;
  size_t sort_temp_bytes;
  unsigned long long cap;
};

static int nms_layout(int64_t n, int64_t edge_capacity, NmsLayout &L) {
  if (edge_capacity <= 0) {
    edge_capacity = 32 * n;
    if (edge_capacity < (1 << 18)) edge_capacity = 1 << 18;
  }
  L.cap = (unsigned long long)edge_capacity;
  size_t tmp = 0;
  uint32_t *nul = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)(n > 0 ? n : 1), 0u, 32u, (hipStream_t)0, false);
  if (e != hipSuccess) return set_err(OBB_EHIP, "rocprim::radix_sort_pairs size query failed: %s", hipGetErrorString(e));
  L.sort_temp_bytes = tmp;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.keys_in = take(nn * 4);
  L.keys_out = take(nn * 4);
  L.vals_in = take(nn * 4);
  L.order = take(nn * 4);
  L.iv = take(nn * sizeof(float4));
  L.rec = take(nn * sizeof(QuadRec));
  L.zero_begin = off;
  L.state = take(nn);
  L.cnt = take(nn * 4);
  L.ctr = take(sizeof(NmsCounters));
  L.zero_end = off;
  L.keepflag = take(nn);
  L.blockcnt = take(((nn + 1023) / 1024) * 4);
  L.sort_temp = take(tmp > 0 ? tmp : 1);
  L.edges = take((size_t)L.cap * sizeof(uint2));
  L.total = off;
  return OBB_OK;
}

}  // namespace obb

extern "C" size_t obb_nms_workspace_bytes(int64_t n, int kind, int64_t edge_capacity) {
  (void)kind;
  obb::NmsLayout L;
  if (n < 0) return 0;
  if (obb::nms_layout(n, edge_capacity, L) != OBB_OK) return 0;
  return L.total;
}

extern "C" int obb_nms_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                           int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                           size_t workspace_bytes, void *stream) {
  using namespace obb;
  if (info_host) { info_host[0] = 0; info_host[1] = 0; info_host[2] = 0; info_host[3] = 0; }
  if (n < 0) return set_err(OBB_EINVAL, "obb_nms_f32: negative n");
  if (kind != OBB_KIND_RBOX && kind != OBB_KIND_POLY) return set_err(OBB_EINVAL, "obb_nms_f32: bad kind %d", kind);
  if (n == 0) return OBB_OK;
  if (n >= (1ll << 31)) return set_err(OBB_EINVAL, "obb_nms_f32: n = %lld exceeds 2^31-1", (long long)n);
  if (!dets_dev || !keep_dev || !workspace_dev) return set_err(OBB_EINVAL, "obb_nms_f32: null pointer");
  if (reinterpret_cast<uintptr_t>(workspace_dev) & 255u) return set_err(OBB_EINVAL, "obb_nms_f32: workspace must be 256-byte aligned");

  // capacity = whatever the caller's workspace holds beyond the fixed part
  NmsLayout L;
  int rc = nms_layout(n, 1, L);
  if (rc != OBB_OK) return rc;
  const size_t fixed = L.edges;
  if (workspace_bytes < fixed + sizeof(uint2))
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: workspace %zu bytes < minimum %zu", workspace_bytes, fixed + sizeof(uint2));
  const unsigned long long cap = (workspace_bytes - fixed) / sizeof(uint2);

  char *ws = static_cast<char *>(workspace_dev);
  uint32_t *keys_in = reinterpret_cast<uint32_t *>(ws + L.keys_in);
  uint32_t *keys_out = reinterpret_cast<uint32_t *>(ws + L.keys_out);
  uint32_t *vals_in = reinterpret_cast<uint32_t *>(ws + L.vals_in);
  uint32_t *order = reinterpret_cast<uint32_t *>(ws + L.order);
  float4 *iv = reinterpret_cast<float4 *>(ws + L.iv);
  QuadRec *rec = reinterpret_cast<QuadRec *>(ws + L.rec);
  uint8_t *state = reinterpret_cast<uint8_t *>(ws + L.state);
  int *cnt = reinterpret_cast<int *>(ws + L.cnt);
  NmsCounters *ctr = reinterpret_cast<NmsCounters *>(ws + L.ctr);
  uint8_t *keepflag = reinterpret_cast<uint8_t *>(ws + L.keepflag);
  int *blockcnt = reinterpret_cast<int *>(ws + L.blockcnt);
  void *sort_temp = ws + L.sort_temp;
  uint2 *edges = reinterpret_cast<uint2 *>(ws + L.edges);
  hipStream_t st = (hipStream_t)stream;

  const int stride = kind == OBB_KIND_POLY ? 9 : 6;
  unsigned gblocks = (unsigned)((n + 255) / 256);
  if (gblocks > 4096) gblocks = 4096;

  NmsProfile &prof = g_prof;
  const bool timed = prof.on;
  if (timed && !prof.created) {
    for (int k = 0; k <= kPhases; ++k) OBB_HIP_TRY(hipEventCreate(&prof.ev[k]));
    prof.created = true;
  }
#define OBB_PHASE(k) do { if (timed) OBB_HIP_TRY(hipEventRecord(prof.ev[k], st)); } while (0)
  OBB_PHASE(0);
  OBB_HIP_TRY(hipMemsetAsync(ws + L.zero_begin, 0, L.zero_end - L.zero_begin, st));
  hipLaunchKernelGGL(nms_key_kernel, dim3(gblocks), dim3(256), 0, st, dets_dev, n, stride, keys_in, vals_in);
  OBB_LAUNCH_CHECK();
  OBB_PHASE(1);
  size_t tmp = L.sort_temp_bytes;
  OBB_HIP_TRY(rocprim::radix_sort_pairs(sort_temp, tmp, keys_in, keys_out, vals_in, order, (size_t)n, 0u, 32u, st, false));
  OBB_PHASE(2);
  if (kind == OBB_KIND_RBOX)
    hipLaunchKernelGGL(nms_gather_kernel<0>, dim3(gblocks), dim3(256), 0, st, dets_dev, n, order, rec, iv);
  else
    hipLaunchKernelGGL(nms_gather_kernel<1>, dim3(gblocks), dim3(256), 0, st, dets_dev, n, order, rec, iv);
  OBB_LAUNCH_CHECK();

  // pair kernel geometry
  const int nj = (int)((n + kNmsColsPerBlock - 1) / kNmsColsPerBlock);
  int rows = kNmsMaxRows;
  auto tiles = [&](int r) {
    const long long G = kNmsColsPerBlock / r;
    return G * ((long long)nj * nj - (long long)nj * (nj - 1) / 2);  // upper bound incl. partial group
  };
  while (rows > 64 && tiles(rows) < 4096) rows >>= 1;
  const long long nblocks = tiles(rows);
  if (nblocks > 2147483647LL) return set_err(OBB_EINVAL, "obb_nms_f32: n = %lld needs too many tiles", (long long)n);
  OBB_PHASE(3);
  hipLaunchKernelGGL(nms_pair_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, iv, rec, (uint32_t)n,
                     iou_thr, rows, nj, edges, cap, ctr, cnt);
  OBB_LAUNCH_CHECK();
  OBB_PHASE(4);
  hipLaunchKernelGGL(nms_resolve_kernel, dim3(1), dim3(1024), 0, st, edges, cap, state, cnt, ctr);
  OBB_LAUNCH_CHECK();
  OBB_PHASE(5);
  hipLaunchKernelGGL(nms_flag_kernel, dim3(gblocks), dim3(256), 0, st, order, state, cnt, n, keepflag);
  OBB_LAUNCH_CHECK();
  const unsigned cblocks = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(nms_count_kernel, dim3(cblocks), dim3(1024), 0, st, keepflag, n, blockcnt);
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(nms_write_kernel, dim3(cblocks), dim3(1024), 0, st, keepflag, n, blockcnt, keep_dev, ctr);
  OBB_LAUNCH_CHECK();

  OBB_PHASE(6);
#undef OBB_PHASE
  NmsCounters h;
  OBB_HIP_TRY(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, st));
  OBB_HIP_TRY(hipStreamSynchronize(st));
  if (timed) {
    for (int k = 0; k < kPhases; ++k) OBB_HIP_TRY(hipEventElapsedTime(&prof.ms[k], prof.ev[k], prof.ev[k + 1]));
    prof.have = true;
  }
  if (info_host) {
    info_host[0] = h.kept;
    info_host[1] = (int64_t)h.edges;
    info_host[2] = h.rounds;
    info_host[3] = (int64_t)cap;
  }
  if (h.edges > cap) {
    if (info_host) info_host[0] = 0;
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: %llu overlap edges exceed the workspace capacity %llu; retry with edge_capacity >= %llu",
                   h.edges, cap, h.edges);
  }
  return OBB_OK;
}

extern "C" int obb_nms_profile_enable(int on) {
  obb::g_prof.on = on != 0;
  obb::g_prof.have = false;
  return OBB_OK;
}

extern "C" int obb_nms_profile_last(float *phase_ms_host) {
  using namespace obb;
  if (!phase_ms_host) return set_err(OBB_EINVAL, "obb_nms_profile_last: null pointer");
  if (!g_prof.have) return set_err(OBB_EINVAL, "obb_nms_profile_last: no profiled obb_nms_f32 call on this thread");
  for (int k = 0; k < kPhases; ++k) phase_ms_host[k] = g_prof.ms[k];
  return OBB_OK;
}
