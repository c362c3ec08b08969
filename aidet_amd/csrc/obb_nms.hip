// obb_nms.hip -- greedy rotated / polygon NMS on gfx950, entirely on the GPU.
//
// Replaces nms_cuda (mmdet/ops/nms/src/nms_kernel.cu:71-139: score sort, dense
// 64x64 bit-mask kernel over all n^2 pairs, 1.25 GB mask D2H copy at n = 100k,
// serial host reduce) with the semantics of the CPU reference
// (mmdet/ops/nms/src/nms_cpu.cpp:21-59).
//
// Greedy NMS is the unique fixpoint of
//     kept(j)  <=>  no kept i that PRECEDES j with overlap(i,j) >= thr
// where "precedes" = higher score, ties -> lower index.  Nothing in that
// definition needs the boxes to be physically sorted by score, so the single
// sort of this pipeline is spent on SPACE instead: boxes are ordered along a
// Morton curve, tiles of 256 consecutive boxes get a bounding box, and only
// (32-row sub-tile, tile) units whose boxes meet are scanned (about 1/70 of all
// pairs at n = 100k, for any distribution -- big boxes just make big tiles).
//
// Three routes (same arithmetic, same edge / resolve formulation):
//   n <= 128      nms_small_kernel: everything in ONE workgroup
//   n <= 10240    direct: prep -> scan of all tile pairs + clip -> resolve + compaction
//   larger        sorted:
//   1 prep     per box: 64-byte quad record, conservative AABB interval, area, score
//              key; per-block bounding box + centre statistics (robust frame);
//              clears the counters of the call
//   2 quant    robust frame, intervals -> 15-bit grid over it, x|y packed in a dword
//              with guard bits; Morton cell of the centre + cell histogram
//   3 sort     counting sort by the Morton CELL of the centre (one scan block, one
//              scatter): only tile locality matters, the order inside a cell is
//              irrelevant for the result
//   4 permute  packed intervals (+ a 16-byte record) in spatial order; bounding box per
//              tile of 256 and per sub-tile of 32
//   5 cull     scan units = (sub-tile of A, tile B > A) whose boxes meet, heavy first
//   6 scan     wave = one unit: only the rows that meet B's box, 3 full-rate VALU ops
//              per pair (rows broadcast with v_readlane, 4 columns per lane), one wave
//              vote per 256 pairs; candidate rows read their per-pair results back and
//              push RAW candidates into a per-wave LDS queue; batches of 64 are pruned
//              (threshold-aware) into sharded global CANDIDATE lists
//   7 clip     one candidate pair per lane, perfectly balanced: the polygon
//              clip (~1100 VALU) runs with every lane busy; pairs with overlap
//              >= thr become directed EDGES (preceding box -> other box) and
//              bump the in-degree word of the target.  The n x n/64 bit matrix of the
//              reference is never materialised.
//   8 resolve  64 workgroups sweep their own edge shards asynchronously: an edge whose
//              source is kept marks its target SUPPRESSED; an edge whose source is
//              suppressed decrements the target's in-degree; a word of zero = kept.
//   9 compact  block scan over the words (already in original index order) ->
//              ascending int64 indices (nms_cpu.cpp:59); counters to pinned host memory.
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include <string.h>

#include "obb_common.h"
#include "obb_geom.h"
#include "obb_nms_internal.h"

namespace obb {

constexpr int kTile = 256;            // boxes per spatial tile
constexpr int kNmsWaves = 4;          // waves per scan block (every wave works on its own units)
constexpr int kNmsColsPerLane = 4;    // 64 lanes x 4 = one tile of columns per wave
constexpr int kSub = 8;               // row sub-tiles per tile
constexpr int kSubRows = kTile / kSub;  // 32 rows: one scan unit = (row sub-tile, column tile)
constexpr int kRowChunk = 4;
constexpr int kNmsQueue = 64 + kTile;  // carry (<= 63) + one row's worst case
constexpr int kCullChunk = 1024;        // column tiles per cull block (LDS list: kSub entries each)
constexpr int kSuper = 16;              // tiles per supertile: the cull kernel tests these first (8 words each)
constexpr int kSuperMinTiles = 1024;    // calls with fewer tiles (n <= 262144) do without supertiles
constexpr int kCells = 262144;          // max Morton cells of the counting sort (9 + 9 bits)
constexpr int kScanBlockCells = 1024;   // cells per block of the multi-block cell scan (kCells / 1024 blocks at most)
constexpr int kShards = kNmsShards;     // candidate lists / counters (atomic rate)
constexpr int kPrepBlocks = 1024;     // upper bound on blocks of the prep kernel
constexpr int kScanBlocks = 2048;     // grid of the scan kernel = one resident set at 8 waves/SIMD (waves loop over units)
// blocks of the clip kernel per candidate shard (lanes loop over their shard's list).  32 was round 2's value: at
// 100k sparse boxes (17 k candidates = 270 per shard) 2048 workgroups were started to find 1 in 30 of them anything to do --
// and STARTING them was 8 of the kernel's 16 us.  8: 11 us there, unchanged on the dense scene (6 100 candidates per
// shard: three rounds per lane, 25 us); 4: 9.5 us / 33 us.
constexpr int kClipBlocksPerShard = 8;

constexpr uint8_t kUndecided = 0, kKept = 1, kSuppressed = 2;
constexpr uint32_t kGuard = 0x80008000u;   // guard bits of the packed intervals
constexpr uint32_t kQMax = 32766u;         // finite boxes quantise to 0..32766
constexpr uint32_t kSentLo = 0x7FFF7FFFu;  // "never a candidate": lo = max, hi = 0
constexpr uint32_t kSentHi = kGuard;

struct alignas(16) QuadRec {  // 64 bytes: four float4 loads
  float ox, oy, ex, ey;
  float rx[4];
  float ry[4];
  float area;
  uint32_t skey;  // score key: smaller = visited earlier by the greedy scan
  int32_t group;  // boxes of different groups never interact (class / image)
  float thr;      // overlap threshold that applies when this box is the earlier one
};

// Hot counters live in separate 256-byte slots: returning atomics on words that
// share a cache line serialise at the memory side.
struct NmsCounters {
  alignas(256) unsigned long long edges;   // overlap edges found (may exceed capacity)
  alignas(256) long long kept;             // K
  long long rounds;                        // resolve sweeps (most of any shard)
  unsigned int done;                       // resolve workgroups finished
  unsigned int state;                      // fused resolve + compaction, after a waiter gave up (flag in `done`): shards that have registered to leave
  alignas(256) unsigned long long cand_total;           // candidate pairs over all shards
  unsigned long long cand_worst;           // fullest shard
};
// What the grid route's INSERT launch counts in.  Apart from NmsCounters since round 6: at a fixed offset of the workspace
// (just below the cell counters), because on a retained workspace (NmsRetain) these three words are cleared by the
// PREVIOUS call's tail -- a call of another n, whose NmsCounters lie elsewhere.
struct NmsGridCtr {
  alignas(256) unsigned int grid_over;     // entries on the overflow list (may exceed grid_over_cap(n))
  unsigned int grid_big;                   // boxes over more than kGridMaxCells cells (may exceed grid_big_cap(n)); (the line of grid_over: both move in rare scenes only)
  alignas(256) unsigned int grid_fail;     // too many crowded cells / a box over too many cells -- repeat on the sorted route
};
constexpr size_t kGridCtrBytes = 1024;     // the slot below the cell counters (cleared with them by the frame kernel)
static_assert(sizeof(NmsGridCtr) <= kGridCtrBytes, "grid counters' slot");

// (NmsHostInfo -- what the host needs after a call, written by the LAST kernel straight into pinned host memory or
// the caller's device record -- lives in obb_nms_internal.h)
static_assert(sizeof(NmsHostInfo) <= OBB_NMS_RESULT_BYTES, "the async result record holds NmsHostInfo");

// narrow-phase flags (kernel arguments; wave-uniform)
constexpr int kFlagStrict = 1;   // suppress on ov > thr (nms_kernel.cu:61) instead of ov >= thr (nms_cpu.cpp:56)
constexpr int kFlagHbb = 2;      // records hold x1,y1,x2,y2 in rx[]: the reference's own pair arithmetic
constexpr int kFlagF64 = 4;      // (with kFlagHbb) the records hold the four coordinates as DOUBLES in rx[] | ry[] and the
                                 // score key has 64 bits (skey | klo[box]): nms_cpu_kernel<double>, nms_cpu.cpp:62-68

// candidate counters: one per shard, each alone in a 256-byte slot (returning
// atomics on words that share a cache line serialise at the memory side)
constexpr int kCandStride = 32;  // in unsigned long long
// scan-unit counters: heavy[kShards] | light[kShards], 128 bytes apart.  The unit list is cut
// into kShards segments (tile A feeds segment A % kShards; heavy units from the front of the
// segment, light ones from its back): ONE pair of counters took a returning atomic from
// every cull block, ~90 per microsecond -- all of the cull kernel's time at 100k boxes.
constexpr int kUnitStride = 32;  // in unsigned int

__device__ __forceinline__ void nms_interval(float o, float e, float &lo, float &hi) {
  // margin 2^-21 (|o| + e): covers every rounding of the spec's prefilter
  float mg = (fabs_sel(o) + e) * 4.76837158203125e-7f;
  lo = (o - e) - mg;
  hi = (o + e) + mg;
}

// order-preserving float <-> uint map (min/max of floats as integers)
__device__ __forceinline__ uint32_t fkey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ uint32_t shfl_xor_u(uint32_t v, int o) {
  return (uint32_t)__shfl_xor((int)v, o);
}

// OBB_KIND_HBB records: the raw x1,y1,x2,y2 travel in rx[] (ry[] unused) so that the narrow
// phase can evaluate nms_cpu.cpp:47-55 on the reference's own floats; origin / extents (the
// theta = 0 box with the +1 widths, load_quad<2>) still feed the broad phase: the continuous
// box [x1-.5, x2+.5] x [y1-.5, y2+.5] meets another one exactly when the reference's
// (xx2 - xx1) + 1 is positive (up to roundings that the interval margin covers).  A box with
// (x2 - x1) + 1 <= 0 or (y2 - y1) + 1 <= 0 has w = 0 or h = 0 against every partner in the
// reference (inter = 0, ovr = 0 or NaN: never >= a positive threshold): it is isolated.
__device__ __forceinline__ void hbb_record(const float *__restrict__ row, QuadRec &r, bool &wellformed) {
  const float x1 = row[0], y1 = row[1], x2 = row[2], y2 = row[3];
  r.rx[0] = x1; r.rx[1] = y1; r.rx[2] = x2; r.rx[3] = y2;
  wellformed = ((x2 - x1) + 1.0f > 0.0f) && ((y2 - y1) + 1.0f > 0.0f);
}

// float64 horizontal boxes (the reference dispatches nms_cpu_kernel on the scalar type, nms_cpu.cpp:62-68, and
// tests/test_nms.py:33-41 runs float64): the broad phase works on the coordinates ROUNDED to fp32 (the interval margin
// of 2^-21 covers the rounding), the narrow phase evaluates nms_cpu.cpp:19,47-56 on the doubles themselves, and
// "precedes" is decided on the 64-bit order-preserving key of the double score.
__device__ __forceinline__ uint64_t score_key64(double s) {
  const double t = s + 0.0;                                     // -0 -> +0
  uint64_t u = (uint64_t)__double_as_longlong(t);
  u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
  return ~u;
}
__device__ __forceinline__ double dmax_sel(double a, double b) { return (a < b) ? b : a; }     // std::max(a, b)
__device__ __forceinline__ double dmin_sel(double a, double b) { return (b < a) ? b : a; }     // std::min(a, b)
__device__ __forceinline__ void quad_doubles(const Quad &q, double (&d)[4]) {
  d[0] = __hiloint2double(__float_as_int(q.rx[1]), __float_as_int(q.rx[0]));
  d[1] = __hiloint2double(__float_as_int(q.rx[3]), __float_as_int(q.rx[2]));
  d[2] = __hiloint2double(__float_as_int(q.ry[1]), __float_as_int(q.ry[0]));
  d[3] = __hiloint2double(__float_as_int(q.ry[3]), __float_as_int(q.ry[2]));
}
__device__ __forceinline__ void rec_put_doubles(QuadRec &r, const double (&d)[4]) {
  r.rx[0] = __int_as_float(__double2loint(d[0])); r.rx[1] = __int_as_float(__double2hiint(d[0]));
  r.rx[2] = __int_as_float(__double2loint(d[1])); r.rx[3] = __int_as_float(__double2hiint(d[1]));
  r.ry[0] = __int_as_float(__double2loint(d[2])); r.ry[1] = __int_as_float(__double2hiint(d[2]));
  r.ry[2] = __int_as_float(__double2loint(d[3])); r.ry[3] = __int_as_float(__double2hiint(d[3]));
}
// nms_cpu.cpp:19,47-56 in double: iarea / area of j, xx1 = max(ix1, x1[j]), ..., w = max(0, xx2 - xx1 + 1), ovr
__device__ __forceinline__ double hbb_overlap_f64(const Quad &A, const Quad &B) {
  double a[4], b[4];
  quad_doubles(A, a); quad_doubles(B, b);
  const double aarea = ((a[2] - a[0]) + 1.0) * ((a[3] - a[1]) + 1.0), barea = ((b[2] - b[0]) + 1.0) * ((b[3] - b[1]) + 1.0);
  const double xx1 = dmax_sel(a[0], b[0]), yy1 = dmax_sel(a[1], b[1]);
  const double xx2 = dmin_sel(a[2], b[2]), yy2 = dmin_sel(a[3], b[3]);
  const double w = dmax_sel(0.0, (xx2 - xx1) + 1.0), h = dmax_sel(0.0, (yy2 - yy1) + 1.0);
  const double inter = w * h;
  return inter / ((aarea + barea) - inter);
}
// one row of doubles -> the fp32 row the broad phase is built from + the record's double payload
template <class Src>
__device__ __forceinline__ void f64_row(const Src &src, int64_t b, float (&rowf)[5], double (&d)[4], uint64_t &key, bool &wellformed) {
  const double *r = src.dets64 + b * src.pitch;
  d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3];
  const double sc = r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rowf[k] = (float)d[k];
  rowf[4] = (float)sc;
  key = score_key64(sc);
  wellformed = ((d[2] - d[0]) + 1.0 > 0.0) && ((d[3] - d[1]) + 1.0 > 0.0);
}

// ---- box source (obb_nms_internal.h): group / threshold of box b, whose row starts at `row`; S = columns of box + score
template <int S>
__device__ __forceinline__ int32_t src_group(const NmsSource &s, int64_t b, const float *__restrict__ row) {
  if (s.mode == 1) return (int32_t)(s.b64[b] * s.mult + s.a64[b]);                 // (wraps like Tensor.to(int32))
  if (s.mode == 2) {
    const int64_t label = (int64_t)row[S], image = (int64_t)row[S + 1];            // (truncation, like Tensor.long())
    return (int32_t)(image * s.mult + label);
  }
  return s.group ? s.group[b] : 0;
}
template <int S>
__device__ __forceinline__ float src_thr(const NmsSource &s, int64_t b, const float *__restrict__ row, float thr) {
  if (s.mode == 2) {
    if (!s.thr_table) return thr;
    const int64_t label = (int64_t)row[S];
    return (label >= 0 && label < s.mult) ? s.thr_table[label] : thr;
  }
  return s.thrs ? s.thrs[b] : thr;
}
__device__ __forceinline__ bool src_grouped(const NmsSource &s) { return s.mode != 0 || s.group != nullptr; }
__device__ __forceinline__ bool src_thresholds(const NmsSource &s) { return s.mode == 2 ? s.thr_table != nullptr : s.thrs != nullptr; }

// ---- 1: prep (DIRECT route: positions are box ids, no global frame, no spatial sort) ----------------------
// per box: 64-byte quad record, conservative float interval, area; clears the call's counters
template <int KIND, bool F64 = false>
__global__ __launch_bounds__(256) void nms_prep_kernel(const NmsSource src, int64_t n,
                                                       QuadRec *__restrict__ rec,
                                                       float4 *__restrict__ iv,
                                                       float *__restrict__ ar,
                                                       float thr, uint4 *__restrict__ zero,
                                                       size_t zero_vecs) {
  constexpr int S = KIND == 0 ? 6 : (KIND == 1 ? 9 : 5);
  // clear the counters / flags of this call (nothing reads them before the next kernel): saves the separate
  // fill launch
  for (size_t z = (size_t)blockIdx.x * blockDim.x + threadIdx.x; z < zero_vecs;
       z += (size_t)gridDim.x * blockDim.x)
    zero[z] = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n;
       b += (int64_t)gridDim.x * blockDim.x) {
    float rowf[5];
    double d64[4];
    uint64_t key64 = 0;
    bool wf64 = true;
    const float *row;
    if constexpr (F64) { f64_row(src, b, rowf, d64, key64, wf64); row = rowf; }
    else row = src.dets + b * src.pitch;
    Quad q;
    load_quad<KIND>(row, q);
    QuadRec r;
    r.ox = q.ox; r.oy = q.oy; r.ex = q.ex; r.ey = q.ey;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.rx[k] = q.rx[k]; r.ry[k] = q.ry[k]; }
    r.area = q.area; r.skey = score_key(row[S - 1]);
    r.group = src_group<S>(src, b, row);
    r.thr = src_thr<S>(src, b, row, thr);
    if (src.gout) src.gout[b] = r.group;
    if (src.tout) src.tout[b] = r.thr;
    bool wellformed = true;
    if (KIND == 2) hbb_record(row, r, wellformed);
    if constexpr (F64) { rec_put_doubles(r, d64); r.skey = (uint32_t)(key64 >> 32); src.klo[b] = (uint32_t)key64; wellformed = wf64; }
    rec[b] = r;
    ar[b] = q.area;
    float4 v;
    nms_interval(q.ox, q.ex, v.x, v.y);
    nms_interval(q.oy, q.ey, v.z, v.w);
    // a box with a non-finite interval can never reach overlap >= thr in the spec
    // (its prefilter / area guards fail): isolate it from the scan
    const float big = 3.0e38f;
    const bool fin = wellformed && (fabs_sel(v.x) < big) && (fabs_sel(v.y) < big) && (fabs_sel(v.z) < big) &&
                     (fabs_sel(v.w) < big);
    if (!fin) { v.x = 1.0f; v.y = -1.0f; v.z = 1.0f; v.w = -1.0f; }  // lo > hi = "isolated"
    iv[b] = v;
  }
}

// ---- 2: quantise + Morton key ----------------------------------------------------
// GROUPED calls (boxes of different groups never interact): the merged NMS of many images
// puts thousands of groups into ONE coordinate range, so a purely spatial order would pair
// every box with its neighbours from all other images (measured: 1e6 boxes of 1000 images
// 9.4 ms against 0.47 ms for one scene of that size).  The sort key therefore starts with a
// hash BUCKET of the group id and only then refines by space: tiles become (nearly) group
// pure, a bucket range per tile / sub-tile culls the units, and the candidate batches drop
// pairs of different groups before they are listed.  The number of bucket bits follows an
// estimate of the number of groups (distinct hashes among 1024 strided samples): few
// groups keep almost all bits for space, thousands of groups give most bits to the bucket.
__device__ __forceinline__ uint32_t group_hash(int32_t g) { return (uint32_t)g * 2654435761u; }
__device__ __forceinline__ uint32_t group_bucket(int32_t g, int gbits) {
  return gbits > 0 ? group_hash(g) >> (32 - gbits) : 0u;
}

__device__ __forceinline__ uint32_t spread15(uint32_t v) {  // 15 bits -> even bit positions
  v &= 0x7FFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

// (floor() of a monotone map is monotone, so lo_a <= hi_b ==> q(lo_a) <= q(hi_b): the quantised test of
// nms_prepq_kernel never loses a real overlap.)

// exclusive scan of the cell counts, in place (one block, PER cells per thread;
// the cell count adapts to n so that this stays a few microseconds)
template <int PER>
__global__ __launch_bounds__(1024) void nms_cellscan_kernel(uint32_t *__restrict__ hist) {
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t v[PER];
  uint32_t tot = 0;
  uint4 *h4 = reinterpret_cast<uint4 *>(hist + tid * PER);
#pragma unroll
  for (int k = 0; k < PER / 4; ++k) {
    const uint4 a = h4[k];
    v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
    tot += a.x + a.y + a.z + a.w;
  }
  uint32_t inc = tot;  // inclusive scan across the wave
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = inc - tot;
  for (int k = 0; k < wave; ++k) base += wsum[k];
#pragma unroll
  for (int k = 0; k < PER / 4; ++k) {
    uint4 o;
    o.x = base; base += v[4 * k];
    o.y = base; base += v[4 * k + 1];
    o.z = base; base += v[4 * k + 2];
    o.w = base; base += v[4 * k + 3];
    h4[k] = o;
  }
}

// sid[position] = box id; positions inside a cell are handed out atomically
// More than 65 536 cells (n > 10^6, or large grouped calls): one block would need a second
// pass over 1 MB; instead 256-thread blocks of 1024 cells chain their totals the way
// nms_compact_kernel does (ticket = rank in start order, a flagged word per block, every block
// sums the words of all earlier tickets).  flags[0] = ticket counter, flags[1 + t] = total | VALID.
__global__ __launch_bounds__(256) void nms_cellscan_big_kernel(uint32_t *__restrict__ hist,
                                                               unsigned int *__restrict__ flags) {
  __shared__ uint32_t wsum[4];
  __shared__ unsigned int s_ticket;
  __shared__ unsigned long long red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(&flags[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned int t = s_ticket;
  uint4 *h4 = reinterpret_cast<uint4 *>(hist + (size_t)t * kScanBlockCells) + tid;
  const uint4 a = *h4;
  const uint32_t tot = a.x + a.y + a.z + a.w;
  uint32_t inc = tot;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = inc - tot, all = 0;
  for (int k = 0; k < 4; ++k) { base += (k < wave) ? wsum[k] : 0u; all += wsum[k]; }
  if (tid == 0) __hip_atomic_store(&flags[1 + t], all | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long part = 0;
  for (unsigned int b = tid; b < t; b += 256) {
    unsigned int v;
    do { v = __hip_atomic_load(&flags[1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v & 0x80000000u));
    part += (unsigned long long)(v & 0x7FFFFFFFu);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  base += (uint32_t)(red[0] + red[1] + red[2] + red[3]);
  uint4 o;
  o.x = base; base += a.x;
  o.y = base; base += a.y;
  o.z = base; base += a.z;
  o.w = base;
  *h4 = o;
}


// =====================================================================================================
// Round 3: the front half of the sorted route in THREE launches instead of six
//   frame     clears the call's counters / histogram / tile-box words (all blocks) and, in block 0, takes the
//             quantisation frame from a strided SAMPLE of <= 4096 boxes (conservative boxes without
//             trigonometry, three rounds of 3-sigma trimming in fp32 about the previous mean).  Any frame
//             gives the same keep set (coordinates clamp into it; DESIGN 3.1 "robust frame") -- only the
//             speed depends on how well it fits, so a sample is as good as the exact bounding box that the
//             prep kernel used to reduce in fp64 over all boxes, and the two kernels that produced and
//             consumed those partials (prep: 11 us, quant: 10.6 us at 100k boxes) become one pass.
//   prepq     per box: quad record, conservative interval -> grid over the frame, Morton cell, histogram
//             add whose returned value is the box's rank inside its cell.
//   sortscat  every block scans the (<= 16384) cell counts into LDS for itself (64 KB from L2: cheaper than
//             a launch boundary), then writes its boxes' packed intervals and 16-byte records straight to
//             their sorted positions -- the cell scan, the index scatter and the gather half of the permute
//             kernel in one launch.
// The tile / sub-tile boxes are then reduced by the CULL kernel itself (nms_cullbox_kernel): a block takes a
// ticket, reduces the boxes of tile `ticket`, publishes the tile box as one 8-byte word and tests its sub-tiles
// against the tiles of all EARLIER tickets, polling their words (they have all started, none waits for anybody
// before publishing: no deadlock; the hand-over pattern of nms_compact_kernel).
// =====================================================================================================
constexpr int kFrameSamples = 1024;     // (4 per thread of ONE 256-thread workgroup: 4096 samples on 16 waves took 14 us, mostly barriers)
#ifndef OBB_FRAME_FEW_LOG2
#define OBB_FRAME_FEW_LOG2 9
#endif
constexpr int kFrameFewLog2 = OBB_FRAME_FEW_LOG2, kFrameFew = 1 << kFrameFewLog2;   // samples of a call below kFrameManyBoxes boxes
constexpr int64_t kFrameManyBoxes = 2000000;   // from here the frame kernel takes all kFrameSamples samples, below half of them
struct alignas(16) NmsFrame { float bx, by, sx, sy; float cpx, cpy; int gbits; int any;
                              int gshx, gshy, ggx, ggy; };   // grid route: cell = quantised coordinate >> gsh, ggx x ggy cells

// conservative axis-aligned box of one input row, no trigonometry (frame only: never affects a result)
template <int KIND>
__device__ __forceinline__ float4 sample_box(const float *__restrict__ row) {
  float lox, hix, loy, hiy;
  if (KIND == 0) {
    const float cx = row[0], cy = row[1], w = row[2], h = row[3];
    const float r = 0.5f * sqrtf(w * w + h * h) * 1.000001f;   // half diagonal >= both half extents
    lox = cx - r; hix = cx + r; loy = cy - r; hiy = cy + r;
  } else if (KIND == 2) {
    lox = row[0] - 0.5f; hix = row[2] + 0.5f; loy = row[1] - 0.5f; hiy = row[3] + 0.5f;
  } else {
    lox = min_sel(min_sel(row[0], row[2]), min_sel(row[4], row[6])); hix = max_sel(max_sel(row[0], row[2]), max_sel(row[4], row[6]));
    loy = min_sel(min_sel(row[1], row[3]), min_sel(row[5], row[7])); hiy = max_sel(max_sel(row[1], row[3]), max_sel(row[5], row[7]));
  }
  const float big = 3.0e38f;
  const bool fin = (fabs_sel(lox) < big) && (fabs_sel(hix) < big) && (fabs_sel(loy) < big) && (fabs_sel(hiy) < big) && lox <= hix && loy <= hiy;
  return fin ? make_float4(lox, hix, loy, hiy) : make_float4(1.0f, -1.0f, 1.0f, -1.0f);
}

// Wave-wide maximum / sum in every lane by DPP moves inside the rows of 16 and four read-lanes (a `__shfl_xor` is a
// ds_bpermute, an LDS-crossbar round trip: the 90 of them in the frame kernel's statistics were 2 us of a one-block chain).
// All 64 lanes must be active.  The sum's order is fixed: the same bits every run.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, dpp_u32<0xB1>(v));     // quad_perm [1,0,3,2]: lane ^ 1
  v = max(v, dpp_u32<0x4E>(v));     // quad_perm [2,3,0,1]: lane ^ 2
  v = max(v, dpp_u32<0x141>(v));    // row_half_mirror
  v = max(v, dpp_u32<0x140>(v));    // row_mirror: every lane holds its row's maximum
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (a + b) + (c + d);
}

// Instrumented build only (-DOBB_FRAME_STAMPS, tools/exp_frame_stamps.py): shader-clock stamps of block 0 of the frame
// kernel (slots 0 .. 7) and wall-clock begin / end of every block of the frame and insert kernels (not part of the ABI).
#ifdef OBB_FRAME_STAMPS
__device__ unsigned long long g_frame_stamps[16 + 2 * 2 * 4096];
#define FR_STAMP(i) do { if (threadIdx.x == 0) g_frame_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FR_WALL(k, e) do { if (threadIdx.x == 0 && blockIdx.x < 4096u) g_frame_stamps[16 + ((k) * 4096 + blockIdx.x) * 2 + (e)] = wall_clock64(); } while (0)
#else
#define FR_STAMP(i) do { } while (0)
#define FR_WALL(k, e) do { } while (0)
#endif
// The frame of a call from a strided sample of its boxes: ONE workgroup of 256 threads (block 0 of nms_frame_kernel; round 6:
// also the extra block of a retained call's insert launch, which takes the frame of THIS call's boxes for the NEXT call).
// `frame2` (may be null) receives a second copy.
template <int KIND, bool F64>
__device__ __forceinline__ void nms_frame_block(const NmsSource &src, int64_t n, int cell_hbits, NmsFrame *__restrict__ frame,
                                                NmsFrame *__restrict__ frame2, int grid_log2) {
  constexpr int S = KIND == 0 ? 6 : (KIND == 1 ? 9 : 5);
  __shared__ float4 sm[kFrameSamples];       // lo x, hi x, lo y, hi y of every sample (lo > hi: not finite)
  __shared__ float red[4][8];
  __shared__ float tot[8];
  __shared__ uint32_t gbm[128];
  constexpr int T = 256, PER = kFrameSamples / T, NW = T / 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // 512 samples below 2 M boxes (round 5: the block's scattered sample reads and its statistics are a chain in front of every
  // call -- 6.1 -> 4.9 us, the kernel 7.2 -> 6.3 us by rocprof), 1024 above (10 M boxes in 240 groups: 1.31 ms against 1.77 --
  // the cell size of a very large scene wants the better estimate)
  const int full = n >= kFrameManyBoxes ? kFrameSamples : kFrameFew;
  const int ns = (int)(n < full ? n : full);
  // sample i sits at box i n / ns: the samples cover [0, n) evenly (a fixed stride n / ns covered [0, ns * stride) only --
  // the first 1024 of 2000 boxes: a scene ordered along x had its second half outside the frame, clamped into the last cells)
  static_assert(kFrameSamples == 1024, "the shifts below");
  auto sample_pos = [&](int i) { return ns == full ? (int64_t)(((uint64_t)i * (uint64_t)n) >> (full == kFrameSamples ? 10 : kFrameFewLog2)) : (int64_t)i; };
  {  // all of a thread's rows are requested before the first is used (a loop of dependent round trips took 30 us)
    float4 b[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * T;
      if constexpr (F64) {
        float rowf[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < ns) { const double *r64 = src.dets64 + sample_pos(i) * src.pitch; for (int k = 0; k < 4; ++k) rowf[k] = (float)r64[k]; }
        b[u] = i < ns ? sample_box<KIND>(rowf) : make_float4(1.0f, -1.0f, 1.0f, -1.0f);
      } else {
        b[u] = i < ns ? sample_box<KIND>(src.dets + sample_pos(i) * src.pitch) : make_float4(1.0f, -1.0f, 1.0f, -1.0f);
      }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) sm[tid + u * T] = b[u];
  }
  // the group ids of <= 1024 strided samples (bucket bits of a grouped call, below) are requested HERE, with the box
  // samples, and used after the frame: at the end of the kernel they were one more exposed round trip per thread
  // (grouped calls: 12 instead of 7.5 us)
  const bool grouped = src_grouped(src);
  uint32_t ghash[4] = {0u, 0u, 0u, 0u};
  const int64_t gs = n < 1024 ? n : 1024;
  if (grouped) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t kk = tid + u * T;
      // (sample kk sits in the kk-th of gs equal strides, at a position inside it that cycles with period 13 -- not a fixed
      // stride, which a class-major or patch-major input may resonate with; no 64-bit division: four of them per thread
      // were ~1.5 us of this one-block kernel)
      if (kk < gs) {
        const uint32_t gstride = (uint32_t)n / (uint32_t)gs;             // (n <= 2^24: 32-bit arithmetic)
        int64_t bb = (int64_t)((uint32_t)kk * gstride + (((uint32_t)kk % 13u) * gstride) / 13u);
        bb = bb < n ? bb : n - 1;
        ghash[u] = group_hash(src_group<S>(src, bb, src.dets + bb * src.pitch)) >> 20;
      }
    }
  }
  if (tid < 128) gbm[tid] = 0u;
  __syncthreads();
  FR_STAMP(2);
  // block-wide sums of up to 8 floats -> tot[], valid after the call
  auto block_sum = [&](float (&v)[8], int m) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = wave_sum_f32(v[k]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) red[wv][k] = v[k];
    }
    __syncthreads();
    if (tid < m) {
      float a = 0.0f;
      for (int w2 = 0; w2 < NW; ++w2) a += red[w2][tid];
      tot[tid] = a;
    }
    __syncthreads();
  };
  __shared__ uint4 wred[NW];
  __shared__ uint4 bbox_all;
  // ---- the usual case in ONE pass: centre statistics about the first sample (fp32: the shift keeps a scene far from
  // the origin from cancelling) and the bounding box of all finite samples.  No centre further than 3.5 sigma from
  // the mean (a uniform scene spans 1.7 sigma) = no outlier: the bounding box is the frame.
  float mx = 0.0f, my = 0.0f, wx = 3.0e38f, wy = 3.0e38f;     // window = mean +- (wx, wy)
  float cnt_in = 0.0f;
  float ext_mean = 0.0f, ext_sd = 0.0f;      // box extent (larger side of the conservative AABB) over the finite samples
  bool robust = false;
  {
    const float4 b0 = sm[0];
    const float px = b0.x <= b0.y ? 0.5f * (b0.x + b0.y) : 0.0f, py = b0.x <= b0.y ? 0.5f * (b0.z + b0.w) : 0.0f;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t axh = 0, axl = 0, ayh = 0, ayl = 0, cxh = 0, cxl = 0, cyh = 0, cyl = 0;
    for (int i = tid; i < ns; i += T) {
      const float4 b = sm[i];
      if (!(b.x <= b.y)) continue;
      const float cx = 0.5f * (b.x + b.y), cy = 0.5f * (b.z + b.w);
      const float dx = cx - px, dy = cy - py;
      v[0] += 1.0f; v[1] += dx; v[2] += dy; v[3] += dx * dx; v[4] += dy * dy;
      { const float e = max_sel(b.y - b.x, b.w - b.z); v[5] += e; v[6] += e * e; }    // (grid route: cell size from the extents)
      axh = max(axh, fkey(b.y)); axl = max(axl, ~fkey(b.x)); ayh = max(ayh, fkey(b.w)); ayl = max(ayl, ~fkey(b.z));
      cxh = max(cxh, fkey(cx)); cxl = max(cxl, ~fkey(cx)); cyh = max(cyh, fkey(cy)); cyl = max(cyl, ~fkey(cy));
    }
    axh = wave_max_u32(axh); axl = wave_max_u32(axl); ayh = wave_max_u32(ayh); ayl = wave_max_u32(ayl);
    cxh = wave_max_u32(cxh); cxl = wave_max_u32(cxl); cyh = wave_max_u32(cyh); cyl = wave_max_u32(cyl);
    __shared__ uint4 cred[NW];
    if (lane == 0) { wred[wv] = make_uint4(axh, axl, ayh, ayl); cred[wv] = make_uint4(cxh, cxl, cyh, cyl); }
    block_sum(v, 7);   // (its barriers also publish wred / cred)
    uint4 m = wred[0], c = cred[0];
    for (int k = 1; k < NW; ++k) {
      m.x = max(m.x, wred[k].x); m.y = max(m.y, wred[k].y); m.z = max(m.z, wred[k].z); m.w = max(m.w, wred[k].w);
      c.x = max(c.x, cred[k].x); c.y = max(c.y, cred[k].y); c.z = max(c.z, cred[k].z); c.w = max(c.w, cred[k].w);
    }
    if (tid == 0) bbox_all = m;
    const float cnt = tot[0];
    if (cnt >= 1.0f) { ext_mean = tot[5] / cnt; const float ev = tot[6] / cnt - ext_mean * ext_mean; ext_sd = ev > 0.0f ? sqrtf(ev) : 0.0f; }
    if (cnt >= 1.0f) {
      const float ex = tot[1] / cnt, ey = tot[2] / cnt;
      float vx = tot[3] / cnt - ex * ex, vy = tot[4] / cnt - ey * ey;
      vx = vx > 0.0f ? vx : 0.0f; vy = vy > 0.0f ? vy : 0.0f;
      mx = px + ex; my = py + ey;
      const float rx = 3.5f * sqrtf(vx) * 1.000001f + 1e-30f + 1e-5f * (fabs_sel(mx) + fabs_sel(px));
      const float ry = 3.5f * sqrtf(vy) * 1.000001f + 1e-30f + 1e-5f * (fabs_sel(my) + fabs_sel(py));
      // any centre beyond the window?  (block-uniform; a non-finite variance -- overflow -- also lands here)
      robust = !(fkey_inv(c.x) - mx <= rx && mx - fkey_inv(~c.y) <= rx && fkey_inv(c.z) - my <= ry && my - fkey_inv(~c.w) <= ry);
      cnt_in = cnt;
    }
    __syncthreads();
  }
  FR_STAMP(3);
  // ---- outliers: three rounds of 3-sigma trimming about the previous mean (round 0: the plain mean, in case the
  // first sample itself was the outlier), then a generous last window (8 sigma: intervals beyond it just clamp)
  if (robust) {
    mx = 0.0f; my = 0.0f; wx = 3.0e38f; wy = 3.0e38f;
    for (int round = 0; round < 4; ++round) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int i = tid; i < ns; i += T) {
        const float4 b = sm[i];
        if (!(b.x <= b.y)) continue;
        const float dx = 0.5f * (b.x + b.y) - mx, dy = 0.5f * (b.z + b.w) - my;
        if (fabs_sel(dx) <= wx && fabs_sel(dy) <= wy) { v[0] += 1.0f; v[1] += dx; v[2] += dy; v[3] += dx * dx; v[4] += dy * dy; }
      }
      block_sum(v, 5);
      const float c = tot[0];
      if (!(c >= 1.0f)) break;                    // (block-uniform) nothing inside: keep the previous window
      cnt_in = c;
      const float ex = tot[1] / c, ey = tot[2] / c;
      float vx = tot[3] / c - ex * ex, vy = tot[4] / c - ey * ey;
      vx = vx > 0.0f ? vx : 0.0f; vy = vy > 0.0f ? vy : 0.0f;
      mx += ex; my += ey;
      const float k = round < 3 ? 3.0f : 8.0f;
      if (round > 0) {
        wx = k * sqrtf(vx) * 1.000001f + 1e-30f + 1e-6f * fabs_sel(mx);
        wy = k * sqrtf(vy) * 1.000001f + 1e-30f + 1e-6f * fabs_sel(my);
      }
      __syncthreads();
    }
  }
  FR_STAMP(4);
  // frame = bounding box of the samples whose centre lies inside the last window (all finite samples when there was
  // no outlier, or when the window leaves too few: a multi-modal scene keeps its bounding box)
  const bool trim = robust && cnt_in * 8.0f >= (float)ns;
  if (trim) {    // (block-uniform)
    uint32_t axh = 0, axl = 0, ayh = 0, ayl = 0;
    for (int i = tid; i < ns; i += T) {
      const float4 b = sm[i];
      if (!(b.x <= b.y)) continue;
      const float dx = 0.5f * (b.x + b.y) - mx, dy = 0.5f * (b.z + b.w) - my;
      if (fabs_sel(dx) <= wx && fabs_sel(dy) <= wy) {
        axh = max(axh, fkey(b.y)); axl = max(axl, ~fkey(b.x));
        ayh = max(ayh, fkey(b.w)); ayl = max(ayl, ~fkey(b.z));
      }
    }
    axh = wave_max_u32(axh); axl = wave_max_u32(axl); ayh = wave_max_u32(ayh); ayl = wave_max_u32(ayl);
    if (lane == 0) wred[wv] = make_uint4(axh, axl, ayh, ayl);
  }
  // bucket bits of a grouped call: distinct 12-bit group hashes among <= 1024 strided samples
  if (grouped) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (tid + u * T < gs) atomicOr(&gbm[ghash[u] >> 5], 1u << (ghash[u] & 31u));
  }
  __syncthreads();
  // distinct 12-bit group hashes among the samples: 128 words counted by two waves (one thread walking them was 128
  // dependent LDS reads, 3-4 us of a grouped call's frame kernel)
  __shared__ int s_distinct;
  if (grouped) {
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    if (tid < 128) {
      int c = __popc(gbm[tid]);
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if (lane == 0) atomicAdd(&s_distinct, c);
    }
    __syncthreads();
  }
  FR_STAMP(5);
  if (tid == 0) {
    uint4 m = bbox_all;
    if (trim) {
      m = wred[0];
      for (int k = 1; k < NW; ++k) {
        m.x = max(m.x, wred[k].x); m.y = max(m.y, wred[k].y);
        m.z = max(m.z, wred[k].z); m.w = max(m.w, wred[k].w);
      }
    }
    NmsFrame f;
    f.any = m.x != 0u;
    float x0 = 0.0f, x1 = 0.0f, y0 = 0.0f, y1 = 0.0f;
    if (f.any) { x0 = fkey_inv(~m.y); x1 = fkey_inv(m.x); y0 = fkey_inv(~m.w); y1 = fkey_inv(m.z); }
    // the sample misses the outermost boxes by ~range / samples: a margin of 8 such steps (cells are 1 / 32767)
    const float gx = (x1 - x0) * (8.0f / (float)full), gy = (y1 - y0) * (8.0f / (float)full);
    x0 -= gx; x1 += gx; y0 -= gy; y1 += gy;
    const float rx = f.any ? x1 - x0 : 0.0f, ry = f.any ? y1 - y0 : 0.0f;
    const bool okx = rx > 0.0f && rx < 3.0e38f, oky = ry > 0.0f && ry < 3.0e38f;
    f.bx = f.any ? x0 : 0.0f; f.by = f.any ? y0 : 0.0f;
    f.sx = okx ? (float)kQMax / rx : 0.0f;
    f.sy = oky ? (float)kQMax / ry : 0.0f;
    // pixels per cell, rounded UP (used for an upper bound of AABB overlaps)
    f.cpx = okx ? (rx / (float)kQMax) * 1.000001f : 0.0f;
    f.cpy = oky ? (ry / (float)kQMax) * 1.000001f : 0.0f;
    int gbits = 0;
    if (grouped) {
      const int d = s_distinct;
      const int total = 2 * cell_hbits;
      int want = 0;
      while ((1 << want) < d) ++want;
      want = (d * 10 >= (int)gs * 7) ? total : want + 2;   // samples nearly all distinct: "very many groups"
      gbits = d <= 1 ? 0 : (want > total - 4 ? total - 4 : want);
    }
    f.gbits = gbits;
    // grid route (nms_ginsert_kernel / nms_gprobe_kernel): cells = the 15-bit quantisation coarsened by a power of two,
    // about (mean + one sigma of the sampled extents) wide -- most boxes touch 1-4 cells -- and at most 2^grid_log2 per axis
    {
#ifndef OBB_GRID_MANY_SCALE
#define OBB_GRID_MANY_SCALE 2.0f
#endif
      // (very many groups -- the per-patch NMS of a tiled pass: ~50 boxes per group -- leave a (group, cell) key nearly
      // empty: cells twice as wide mean 1.5 instead of 1.9 entries per box and fewer pairs per table cell -- the
      // per-patch phase of config 5 277 -> 234 us; three times as wide: a (group, cell) key overflows, the route gives up)
      const bool many_groups = grouped && s_distinct * 10 >= (int)gs * 7;
      const float ctarget = (ext_mean + ext_sd) * (many_groups ? OBB_GRID_MANY_SCALE : 1.0f);    // (half that: probe 17 -> 25 us at the headline, merged NMS of config 5 440 -> 585 us; LOG)
      auto shift_for = [&](float scale) {
        const float q = ctarget * scale;                      // cell size in quanta
        int sh = 15 - grid_log2;
        while (sh < 15 && (float)(1 << sh) < q) ++sh;
        return sh;
      };
      f.gshx = shift_for(f.sx); f.gshy = shift_for(f.sy);
      f.ggx = (int)(kQMax >> f.gshx) + 1; f.ggy = (int)(kQMax >> f.gshy) + 1;
    }
    *frame = f;
    if (frame2) *frame2 = f;
  }
  FR_STAMP(6);
}

template <int KIND, bool F64 = false>
__global__ __launch_bounds__(256) void nms_frame_kernel(const NmsSource src, int64_t n, int cell_hbits,
                                                        NmsFrame *__restrict__ frame, NmsFrame *__restrict__ frame2,
                                                        uint4 *__restrict__ zero, size_t zero_vecs,
                                                        uint4 *__restrict__ zero2, size_t zero2_vecs, int grid_log2) {
  FR_WALL(0, 0);
  if (blockIdx.x == 0) FR_STAMP(0);
  for (size_t z = (size_t)blockIdx.x * blockDim.x + threadIdx.x; z < zero_vecs; z += (size_t)gridDim.x * blockDim.x)
    zero[z] = make_uint4(0u, 0u, 0u, 0u);
  // (grid route: the cell counters, a range of their own since round 6 -- at a fixed offset of the workspace)
  for (size_t z = (size_t)blockIdx.x * blockDim.x + threadIdx.x; z < zero2_vecs; z += (size_t)gridDim.x * blockDim.x)
    zero2[z] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x != 0) { FR_WALL(0, 1); return; }
  FR_STAMP(1);
  nms_frame_block<KIND, F64>(src, n, cell_hbits, frame, frame2, grid_log2);
  FR_WALL(0, 1);
}

#ifdef OBB_FRAME_STAMPS
}  // namespace obb
extern "C" __attribute__((visibility("default"))) int obb_debug_frame_stamps(unsigned long long *host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_frame_stamps), sizeof(unsigned long long) * (size_t)words);
}
namespace obb {
#endif

template <int KIND, bool F64 = false>
__global__ __launch_bounds__(256) void nms_prepq_kernel(const NmsSource src, int64_t n,
                                                        QuadRec *__restrict__ rec, uint2 *__restrict__ qb,
                                                        float *__restrict__ ar, uint32_t *__restrict__ ckey,
                                                        uint32_t *__restrict__ crank, uint32_t *__restrict__ hist,
                                                        int cell_hbits, const NmsFrame *__restrict__ frame, float thr) {
  constexpr int S = KIND == 0 ? 6 : (KIND == 1 ? 9 : 5);
  const NmsFrame fr = *frame;
  const int gbits = fr.gbits, cbits = 2 * cell_hbits - gbits;
  auto qz = [](float x, float b, float s) {
    float u = floorf((x - b) * s);
    u = u < 0.0f ? 0.0f : (u > (float)kQMax ? (float)kQMax : u);   // (NaN -> kQMax by the second select: never happens, x is finite)
    return (uint32_t)u;
  };
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n; b += (int64_t)gridDim.x * blockDim.x) {
    float rowf[5];
    double d64[4];
    uint64_t key64 = 0;
    bool wf64 = true;
    const float *row;
    if constexpr (F64) { f64_row(src, b, rowf, d64, key64, wf64); row = rowf; }
    else row = src.dets + b * src.pitch;
    Quad q;
    load_quad<KIND>(row, q);
    QuadRec r;
    r.ox = q.ox; r.oy = q.oy; r.ex = q.ex; r.ey = q.ey;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.rx[k] = q.rx[k]; r.ry[k] = q.ry[k]; }
    r.area = q.area; r.skey = score_key(row[S - 1]);
    const int32_t g = src_group<S>(src, b, row);
    r.group = g;
    r.thr = src_thr<S>(src, b, row, thr);
    if (src.gout) src.gout[b] = g;
    if (src.tout) src.tout[b] = r.thr;
    bool wellformed = true;
    if (KIND == 2) hbb_record(row, r, wellformed);
    if constexpr (F64) { rec_put_doubles(r, d64); r.skey = (uint32_t)(key64 >> 32); src.klo[b] = (uint32_t)key64; wellformed = wf64; }
    rec[b] = r;
    ar[b] = q.area;
    float4 v;
    nms_interval(q.ox, q.ex, v.x, v.y);
    nms_interval(q.oy, q.ey, v.z, v.w);
    // a box with a non-finite interval can never reach overlap >= thr in the spec (its prefilter / area
    // guards fail): isolate it from the scan
    const float big = 3.0e38f;
    const bool fin = wellformed && (fabs_sel(v.x) < big) && (fabs_sel(v.y) < big) && (fabs_sel(v.z) < big) && (fabs_sel(v.w) < big);
    uint2 o = make_uint2(kSentLo, kSentHi);
    uint32_t mk = (1u << (2 * cell_hbits)) - 1u;  // isolated boxes go to the last cell
    if (fin) {
      const uint32_t xl = qz(v.x, fr.bx, fr.sx), xh = qz(v.y, fr.bx, fr.sx);
      const uint32_t yl = qz(v.z, fr.by, fr.sy), yh = qz(v.w, fr.by, fr.sy);
      o.x = xl | (yl << 16);
      o.y = (xh | (yh << 16)) | kGuard;
      const int sh = 16 - cell_hbits;  // (lo + hi) is a 16-bit centre*2: keep its top bits
      mk = spread15((xl + xh) >> sh) | (spread15((yl + yh) >> sh) << 1);
      if (gbits > 0) mk = (group_bucket(g, gbits) << cbits) | (mk >> gbits);   // bucket, then coarser space
    }
    qb[b] = o;
    ckey[b] = mk;
    // the count of the cell AND this box's rank inside it (one round of scattered atomics per call).
    // Grouped calls: the rows of a tiled pass arrive in patch order, so the lanes of a wave share cells, and the merged
    // rows put ~40 boxes into every used cell (30 groups hash into 30 of 128 buckets): returning atomics on ONE word
    // serialise at the memory side (26 us for 156k merged rows against 11 us for as many per-patch rows,
    // tools/exp_merge_prepq.py).  The lanes of a wave that share a cell send ONE atomic (one round per distinct cell
    // in the wave) and rank themselves behind its result.
    if (gbits > 0) {
      const int lane = threadIdx.x & 63;
      uint64_t rem = __ballot(true);
      uint32_t myrank = 0, mycnt = 1;
      int mylead = lane;
      while (rem) {
        const int lead = __builtin_ctzll(rem);
        const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)mk, lead);
        const uint64_t m = __ballot(mk == key);
        if (mk == key) { myrank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); mycnt = (uint32_t)__popcll(m); mylead = lead; }
        rem &= ~m;
      }
      uint32_t base = 0;
      if (mylead == lane) base = atomicAdd(&hist[mk], mycnt);
      base = (uint32_t)__shfl((int)base, mylead);
      crank[b] = base + myrank;
    } else {
      crank[b] = atomicAdd(&hist[mk], 1u);
    }
  }
}

// =====================================================================================================================
// GRID route (round 4): frame -> insert -> probe + clip -> resolve + compaction = FOUR launches for ungrouped calls of
// 10 241 .. 262 143 boxes whose scene is sparse enough (the headline's is: 100 000 boxes of 8-128 px on a 20 000^2
// canvas).  The sorted route pays four launches (frame, prepq, sort-scatter, cull-box) before its first pair test to
// be distribution-proof; a uniform grid needs none of that when no cell is crowded:
//   insert   per box: the 64-byte record (as nms_prep_kernel), the packed 15-bit interval, and one 16-byte entry
//            (interval, box id, area) into every cell its interval touches -- cells = the quantisation coarsened by a power of
//            two, about one mean + sigma extent wide (nms_frame_kernel), fixed capacity kGridCap per cell;
//   probe    a wave per chunk of 2-8 cells, a lane per pair of entries of one cell (nms_gprobe_kernel); a pair whose
//            intervals meet is a candidate IN ONE CELL ONLY -- the one that holds the lower corner of the intersection of
//            the two intervals (both boxes touch it: the cell function is monotone) -- and goes through the same per-wave
//            queue, pruning and clip as the direct route (DirectSink::flush); the edges are what every route produces;
//   resolve  nms_resolve_kernel<false> with its fused compaction.
// A crowded cell (more than kGridCap entries) or a box that touches more than kGridMaxCells cells raises `grid_fail`:
// the call's record says route 3, the synchronous entry points repeat the call on the sorted route and the calling
// thread remembers it for the size class (the next kGridSkipCalls calls go straight to the sorted route).
// =====================================================================================================================
constexpr int kGridCap = 32;             // entries per cell
constexpr int kGridMaxCells = 64;        // cells one box may touch
constexpr int kGridMaxLog2 = 11;         // at most 2048 x 2048 cells per group
constexpr int kGridMaxBits = 22;         // at most 2^22 cells in the table (2 GiB: calls of ~10^7 boxes)
constexpr int kGridOverMax = 49152;      // the overflow list's memory (entries that found their cell full) ...
// ... of which a call uses max(4096, n / 256): more entries and the route gives up (every entry costs a wave a pass over the
// list: 4096 entries ~10 us, 32 768 ~0.5 ms -- in proportion only for the calls of millions of boxes that may have them;
// the merged NMS of 10^7 tiled detections puts ~32 000 there)
// boxes that touch more than kGridMaxCells cells (an airport among vehicles) are not inserted: they go on the BIG list and
// the probe tests each of them against every box of the call (20 bytes per box: its record's intervals, its area); the list holds
// 1024 boxes and at most 2^28 / n (1.25 M boxes: 214), more and the route gives up
constexpr int kGridBigMax = 1024;
__host__ __device__ inline uint32_t grid_big_cap(int64_t n) {
  const int64_t c = ((int64_t)1 << 28) / (n > 0 ? n : 1);
  return (uint32_t)(c > kGridBigMax ? kGridBigMax : (c < 1 ? 1 : c));
}
__host__ __device__ inline uint32_t grid_over_cap(int64_t n) {
  const int64_t c = n / 256;
  return (uint32_t)(c < 4096 ? 4096 : (c > kGridOverMax ? kGridOverMax : c));
}
constexpr int kGridSkipCalls = 256;
constexpr int64_t kGridAutoMax = 12000000;   // (the layout holds a table up to 2^24 boxes)

// ---- RETAINED workspace (round 6; OBB_NMS_RETAIN): three launches instead of four ------------------------------------
// The frame kernel has two jobs in front of every call -- clearing what the call's atomics count in, and the frame -- and
// neither needs THIS call's boxes: any frame gives the same keep set (coordinates clamp into it monotonically, pairs are
// only ever added), and the cell counters are zero again as soon as somebody clears them.  A caller that hands the SAME
// workspace to call after call and promises not to write it in between lets the library keep both in it:
//   * the tail of a call (the resolve launch, after the probe has read them) clears the cell counters and the three
//     words the insert launch counts in (grid_over, grid_big, grid_fail); everything the LATER launches of a call count in
//     (in-degree words, candidate / edge counters, the compaction's hand-over words) is cleared by the insert launch itself;
//   * an extra workgroup of the insert launch takes the frame of THIS call's boxes (nms_frame_block: the frame kernel's
//     block 0) and leaves it for the NEXT call, which reads the frame its predecessor left: a stream of scenes of one
//     kind (the tiles of an image, the frames of a video, the bench's pool) gets frames as good as its own, a scene unlike
//     its predecessor at worst a crowded grid -- the call is then repeated with the frame kernel in front (seeding).
// The header at offset 0 of the workspace names the call the state is ready for (magic = f(sequence number, layout
// signature)); the host keeps the expected value per workspace pointer (thread-local) and every launch of a retained call
// checks it first: a workspace that somebody else wrote (or another thread used) makes the call leave without touching
// anything and the host seeds it again.  Applies to the grid route below 2^18 boxes (the fused compaction is the tail).
struct alignas(64) NmsRetain {
  unsigned long long magic;
  unsigned long long pad;
  NmsFrame frames[2];          // call number q reads frames[q & 1] and leaves the frame of its own boxes in frames[(q + 1) & 1]
};
static_assert(sizeof(NmsRetain) <= 256, "the header's slot of the layout");
constexpr unsigned long long kRetainMagic = 0x6F62626E6D733036ull;
__host__ __device__ inline unsigned long long retain_magic(unsigned long long seq, unsigned int sig) {
  return kRetainMagic ^ (seq * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)sig << 20);
}
// the layout signature the header's magic carries: table geometry (what the tail clears, what a frame's grid shifts assume)
static inline unsigned int retain_sig(int grid_log2, int grid_bits, bool grouped) {
  return (unsigned int)grid_log2 | ((unsigned int)grid_bits << 8) | (grouped ? 1u << 16 : 0u);
}
// what a retained call's launches need beyond the plain grid route's arguments (by value; all zero: a plain call)
struct RetainArgs {
  const unsigned long long *hdr_magic;   // header word to check against `expect` (null: no check -- a seeding call)
  unsigned long long expect;
  NmsFrame *frame_next;                  // insert launch: where the extra workgroup leaves the frame of this call's boxes
  int *cnt;                              // insert launch: in-degree words, cleared per box
  unsigned long long *ccnts;             // ... the candidate / edge counters
  unsigned int *ucnts, *blockcnt;        // ... scan-unit counters (read by the record), hand-over words of the fused compaction
};
struct RetainTail {
  unsigned long long *hdr_magic;         // null: no tail (a plain call)
  unsigned long long expect, next;       // expect != 0: check first; next: the header's value once the state is ready again
  uint4 *gcnt;                           // cell counters: every workgroup of the resolve launch clears its share as the last thing it does
  size_t gcnt_vecs;
};
constexpr unsigned int kRouteUnseeded = 6u;   // record of a retained call that found the header changed: nothing was computed

// GROUPED calls share the table: a group's cells are the spatial grid shifted by a pseudo-random offset that depends on a
// 16-bit tag of the group id, modulo the table size (a power of two >= the spatial grid: one group's cells never alias
// each other).  The entry carries the tag: entries of different tags are never paired; two groups of one tag (or that
// meet in a cell) are told apart by the records, as everywhere (DirectSink::flush).  Ungrouped: tag 0, offset 0.
__device__ __forceinline__ uint32_t grid_tag(int32_t group) { return group_hash(group) >> 16; }
__device__ __forceinline__ uint32_t grid_tag_offset(uint32_t tag, uint32_t mask) { return (tag * 0x9E3779B1u >> 8) & mask; }
// the entry's last word: the upper half of the area's float bits (a lower bound of the area, within 2^-7 of it) | tag
__device__ __forceinline__ uint32_t grid_entry_word(float area, uint32_t tag) { return (__float_as_uint(area) & 0xFFFF0000u) | tag; }

// (ablation builds for tools/exp: OBB_GINS_ABL 1 = no table stores, 2 = no atomics, 3 = neither)
#ifndef OBB_GINS_ABL
#define OBB_GINS_ABL 0
#endif
#if OBB_GINS_ABL & 2
#define GINS_ATOMIC(cnt_, b_, cell_) ((uint32_t)(b_) & 31u)
#else
#define GINS_ATOMIC(cnt_, b_, cell_) atomicAdd(&(cnt_)[cell_], 1u)
#endif
#if OBB_GINS_ABL & 1
#define GINS_STORE(tab_, cell_, slot_, e_) do { if ((cell_) == 0xFFFFFFFFu) (tab_)[0] = (e_); } while (0)
#else
#define GINS_STORE(tab_, cell_, slot_, e_) (tab_)[(size_t)(cell_) * kGridCap + (slot_)] = (e_)
#endif
// a box's packed 15-bit interval in the call's frame and the cells it touches (shared by the insert kernel and the big
// boxes' pass of the probe, which rebuilds it from the record: the same expressions, the same bits).  false: isolated.
__device__ __forceinline__ bool grid_interval(float ox, float ex, float oy, float ey, bool wellformed, const NmsFrame &fr,
                                              uint2 &o, uint32_t &cx0, uint32_t &cx1, uint32_t &cy0, uint32_t &cy1) {
  float4 v;
  nms_interval(ox, ex, v.x, v.y);
  nms_interval(oy, ey, v.z, v.w);
  const float big = 3.0e38f;
  const bool fin = wellformed && (fabs_sel(v.x) < big) && (fabs_sel(v.y) < big) && (fabs_sel(v.z) < big) && (fabs_sel(v.w) < big);
  o = make_uint2(kSentLo, kSentHi);             // an isolated box: in no cell, meets nothing
  cx0 = cx1 = cy0 = cy1 = 0u;
  if (!fin) return false;
  auto qz = [](float x, float b, float s2) {
    float u = floorf((x - b) * s2);
    u = u < 0.0f ? 0.0f : (u > (float)kQMax ? (float)kQMax : u);
    return (uint32_t)u;
  };
  const uint32_t xl = qz(v.x, fr.bx, fr.sx), xh = qz(v.y, fr.bx, fr.sx);
  const uint32_t yl = qz(v.z, fr.by, fr.sy), yh = qz(v.w, fr.by, fr.sy);
  o.x = xl | (yl << 16);
  o.y = (xh | (yh << 16)) | kGuard;
  cx0 = xl >> fr.gshx; cx1 = xh >> fr.gshx; cy0 = yl >> fr.gshy; cy1 = yh >> fr.gshy;
  return true;
}

template <int KIND>
__global__ __launch_bounds__(256) void nms_ginsert_kernel(const NmsSource src, int64_t n, QuadRec *__restrict__ rec,
                                                          float *__restrict__ ar,
                                                          const NmsFrame *__restrict__ frame, float thr,
                                                          uint32_t *__restrict__ gcnt, uint4 *__restrict__ gtab,
                                                          NmsCounters *__restrict__ ctr, NmsGridCtr *__restrict__ gctr, uint32_t mask,
                                                          uint4 *__restrict__ gover, uint32_t *__restrict__ gbig,
                                                          uint32_t big_cap, int give_up,    // (big_cap = grid_big_cap(n), a 64-bit division: the host's)
                                                          const RetainArgs ra, int cell_hbits, int grid_log2) {
  constexpr int S = KIND == 0 ? 6 : (KIND == 1 ? 9 : 5);
  FR_WALL(1, 0);
  // ---- retained call (no frame launch in front; see NmsRetain): the header must name THIS call, the last workgroup takes
  // the frame of this call's boxes for the next one, and what the later launches count in is cleared here
  unsigned int nblk = gridDim.x;
  if (ra.frame_next) {
    if (ra.hdr_magic && *ra.hdr_magic != ra.expect) return;      // (uniform over the launch: nothing is touched)
    nblk = gridDim.x - 1u;
    if (blockIdx.x == nblk) {
#ifndef OBB_INSERT_NOFRAME      // (timing experiment: the launch without the frame block's 17 KB of LDS; the frame of the seeding call stays)
      nms_frame_block<KIND, false>(src, n, cell_hbits, ra.frame_next, nullptr, grid_log2);
#endif
      return;
    }
    if (blockIdx.x == 0) {
      const unsigned int t = threadIdx.x;
      if (t < 3u * kShards) ra.ccnts[(size_t)t * kCandStride] = 0ull;
      if (t < 2u * kShards) ra.ucnts[(size_t)t * kUnitStride] = 0u;
      if (t < (unsigned)kShards + 1u) ra.blockcnt[t] = 0u;
      if (t == 255u) {       // (grid_over, grid_big, grid_fail are this launch's own: the previous call's tail cleared them)
        ctr->edges = 0ull; ctr->kept = 0; ctr->rounds = 0; ctr->done = 0u; ctr->state = 0u; ctr->cand_total = 0ull; ctr->cand_worst = 0ull;
      }
    }
  }
  const NmsFrame fr = *frame;
  uint32_t *__restrict__ gover_cell = reinterpret_cast<uint32_t *>(gover + kGridOverMax);
  const uint32_t over_cap = grid_over_cap(n);
  // an entry whose cell is full goes on the overflow list (nms_gprobe_kernel pairs it with the cell's entries and with the
  // list's other entries of the cell); a full list gives the route up
  auto spill = [&](uint32_t cell, const uint4 &e, uint32_t failword) {
    const uint32_t o = atomicAdd(&gctr->grid_over, 1u);
    if (o < over_cap) { gover[o] = e; gover_cell[o] = cell; }
    else gctr->grid_fail = failword;
  };
  const bool grouped = src_grouped(src);
  const uint32_t fail = 1u + (uint32_t)KIND + (grouped ? 4u : 0u) + 8u * (uint32_t)src.mode;   // (non-zero; names the size class's skip hint: kind, grouped, source mode)
  if (give_up && blockIdx.x == 0 && threadIdx.x == 0) gctr->grid_fail = fail;      // (test hook, OBB_TUNE_NMS_GRID = 2: the repeat paths without a crowded scene)
  const int xbits = 15 - fr.gshx;                  // ggx = 2^xbits
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n; b += (int64_t)nblk * blockDim.x) {
    const float *row = src.dets + b * src.pitch;
    Quad q;
    load_quad<KIND>(row, q);
    QuadRec r;
    r.ox = q.ox; r.oy = q.oy; r.ex = q.ex; r.ey = q.ey;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.rx[k] = q.rx[k]; r.ry[k] = q.ry[k]; }
    r.area = q.area; r.skey = score_key(row[S - 1]);
    r.group = grouped ? src_group<S>(src, b, row) : 0;
    r.thr = src_thr<S>(src, b, row, thr);
    bool wellformed = true;
    if (KIND == 2) hbb_record(row, r, wellformed);
    rec[b] = r;
    ar[b] = q.area;
    if (ra.frame_next) ra.cnt[b] = 0;
    const uint32_t tag = grouped ? grid_tag(r.group) : 0u;
    const uint32_t goff = grid_tag_offset(tag, mask);
    uint2 o;
    uint32_t cx0, cx1, cy0, cy1;
    if (grid_interval(q.ox, q.ex, q.oy, q.ey, wellformed, fr, o, cx0, cx1, cy0, cy1)) {
      const uint32_t ncell = (cx1 - cx0 + 1u) * (cy1 - cy0 + 1u);
      const uint4 ent = make_uint4(o.x, o.y, (uint32_t)b, grid_entry_word(r.area, tag));
      if (ncell > (uint32_t)kGridMaxCells) {
        const uint32_t k = atomicAdd(&gctr->grid_big, 1u);
        if (k < big_cap) gbig[k] = (uint32_t)b;
        else gctr->grid_fail = fail;
      } else if (cx1 - cx0 <= 1u && cy1 - cy0 <= 1u) {
        // the usual box (a cell is about mean + sigma extent wide): 1-4 cells, their slots requested TOGETHER -- a
        // returning device-scope atomic is a round trip of ~2 us and four of them in a row were most of this kernel
        const bool two_x = cx1 > cx0, two_y = cy1 > cy0;
        const uint32_t s00c = ((cy0 << xbits) | cx0) + goff;
        const uint32_t c00 = s00c & mask, c10 = (s00c + 1u) & mask, c01 = (s00c + (uint32_t)fr.ggx) & mask, c11 = (s00c + (uint32_t)fr.ggx + 1u) & mask;
        uint32_t s00 = 0, s10 = 0, s01 = 0, s11 = 0;
#if OBB_GINS_ABL == 0
        if (two_x && (c00 & 1u) == 0u) {
          // x neighbours whose counters share an aligned 8-byte word: ONE 64-bit add serves both (the low counter never
          // carries into the high one) -- a request less at the memory-side atomic units for every second such box
          const unsigned long long r0 = atomicAdd(reinterpret_cast<unsigned long long *>(gcnt + c00), 0x100000001ull);
          s00 = (uint32_t)r0; s10 = (uint32_t)(r0 >> 32);
          if (two_y) {       // (c01 = c00 + ggx has c00's parity: ggx is a power of two >= 2 here)
            const unsigned long long r1 = atomicAdd(reinterpret_cast<unsigned long long *>(gcnt + c01), 0x100000001ull);
            s01 = (uint32_t)r1; s11 = (uint32_t)(r1 >> 32);
          }
        } else
#endif
        {
          s00 = GINS_ATOMIC(gcnt, b, c00);
          if (two_x) s10 = GINS_ATOMIC(gcnt, b, c10);
          if (two_y) s01 = GINS_ATOMIC(gcnt, b, c01);
          if (two_x && two_y) s11 = GINS_ATOMIC(gcnt, b, c11);
        }
        if (s00 < (uint32_t)kGridCap) GINS_STORE(gtab, c00, s00, ent); else spill(c00, ent, fail);
        if (two_x) { if (s10 < (uint32_t)kGridCap) GINS_STORE(gtab, c10, s10, ent); else spill(c10, ent, fail); }
        if (two_y) { if (s01 < (uint32_t)kGridCap) GINS_STORE(gtab, c01, s01, ent); else spill(c01, ent, fail); }
        if (two_x && two_y) { if (s11 < (uint32_t)kGridCap) GINS_STORE(gtab, c11, s11, ent); else spill(c11, ent, fail); }
      } else {
        for (uint32_t cy = cy0; cy <= cy1; ++cy)
          for (uint32_t cx = cx0; cx <= cx1; ++cx) {
            const uint32_t cell = (((cy << xbits) | cx) + goff) & mask;
            const uint32_t slot = GINS_ATOMIC(gcnt, b, cell);
            if (slot < (uint32_t)kGridCap) GINS_STORE(gtab, cell, slot, ent);
            else spill(cell, ent, fail);
          }
      }
    }
  }
  FR_WALL(1, 1);
}

// CELLS > 0: the block scans hist[0 .. CELLS) into LDS itself; CELLS == 0: `hist` already holds the offsets
// (calls with more than 16384 cells run the cell-scan kernels first).  One box per thread and step.
template <int CELLS>
__global__ __launch_bounds__(1024) void nms_sortscatter_kernel(const uint32_t *__restrict__ hist,
                                                               const uint2 *__restrict__ qb,
                                                               const float *__restrict__ ar,
                                                               const uint32_t *__restrict__ ckey,
                                                               const uint32_t *__restrict__ crank, int64_t n,
                                                               int64_t npos, const int32_t *__restrict__ group,
                                                               const float *__restrict__ thrs,
                                                               uint2 *__restrict__ qs, uint4 *__restrict__ pf,
                                                               int32_t *__restrict__ pg, float *__restrict__ pt) {
  __shared__ uint32_t offs[CELLS > 0 ? CELLS : 1];
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * 1024 + tid;
  // this thread's first box is requested before the scan: its round trip overlaps it
  uint32_t key0 = 0, rank0 = 0;
  if (b0 < n) { key0 = ckey[b0]; rank0 = crank[b0]; }
  if (CELLS > 0) {
    constexpr int PER = CELLS > 0 ? CELLS / 1024 : 4;
    uint32_t v[PER], tot = 0;
    const uint4 *h4 = reinterpret_cast<const uint4 *>(hist + tid * PER);
#pragma unroll
    for (int k = 0; k < PER / 4; ++k) {
      const uint4 a = h4[k];
      v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
      tot += a.x + a.y + a.z + a.w;
    }
    uint32_t inc = tot;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - tot;
    for (int k = 0; k < wave; ++k) base += wsum[k];
#pragma unroll
    for (int k = 0; k < PER; ++k) { offs[tid * PER + k] = base; base += v[k]; }
    __syncthreads();
  }
  for (int64_t b = b0; b < n; b += (int64_t)gridDim.x * 1024) {
    const uint32_t key = (b == b0) ? key0 : ckey[b], rank = (b == b0) ? rank0 : crank[b];
    const uint32_t pos = (CELLS > 0 ? offs[key] : hist[key]) + rank;
    const uint2 o = qb[b];
    qs[pos] = o;
    // one 16-byte record per position for the candidate batches (one gather instead of three)
    pf[pos] = make_uint4(o.x, o.y, __float_as_uint(ar[b]), (uint32_t)b);
    if (pg) pg[pos] = group[b];
    if (pt) pt[pos] = thrs[b];
  }
  // the padding of the last tile: sentinels (they meet nothing)
  if (blockIdx.x == 0) {
    for (int64_t s2 = n + tid; s2 < npos; s2 += 1024) {
      qs[s2] = make_uint2(kSentLo, kSentHi);
      pf[s2] = make_uint4(kSentLo, kSentHi, 0u, 0u);
      if (pg) pg[s2] = 0;
      if (pt) pt[s2] = 0.0f;
    }
  }
}

// ---- 4: tile bounding boxes (calls of >= kSuperMinTiles tiles; smaller ones: nms_cullbox_kernel) -------------
// one block = one tile of 256 spatial positions: bounding box per tile and per sub-tile of 32, bucket ranges of
// grouped calls, supertile boxes
__global__ __launch_bounds__(256) void nms_tilebox_kernel(const uint2 *__restrict__ qs, int64_t n,
                                                          uint2 *__restrict__ tbox,
                                                          uint2 *__restrict__ sbox,
                                                          const int32_t *__restrict__ pg,
                                                          const NmsFrame *__restrict__ frame,
                                                          uint32_t *__restrict__ tgr,
                                                          uint32_t *__restrict__ sgr,
                                                          unsigned int *__restrict__ super) {
  __shared__ uint4 wred[4];
  __shared__ uint2 gred[4];
  const int64_t s = (int64_t)blockIdx.x * kTile + threadIdx.x;
  const uint2 o = qs[s];
  // grouped calls: the bucket range (lo | hi << 16) of every sub-tile / tile (empty: lo > hi)
  uint32_t glo = 0xFFFFu, ghi = 0u;
  if (pg) {
    // (padding and isolated boxes meet nothing: they do not widen the range)
    if (s < n && o.x != kSentLo) { glo = ghi = group_bucket(pg[s], frame->gbits); }
  }
  uint32_t xl = o.x & 0x7FFFu, yl = (o.x >> 16) & 0x7FFFu;
  uint32_t xh = o.y & 0x7FFFu, yh = (o.y >> 16) & 0x7FFFu;
  for (int k = kSubRows / 2; k > 0; k >>= 1) {
    xl = min(xl, shfl_xor_u(xl, k)); yl = min(yl, shfl_xor_u(yl, k));
    xh = max(xh, shfl_xor_u(xh, k)); yh = max(yh, shfl_xor_u(yh, k));
  }
  // bounding box of each run of 32 consecutive positions (a sub-tile of sentinels
  // keeps lo = max, hi = 0: it meets nothing)
  if (pg) {
    for (int k = kSubRows / 2; k > 0; k >>= 1) { glo = min(glo, shfl_xor_u(glo, k)); ghi = max(ghi, shfl_xor_u(ghi, k)); }
    if ((threadIdx.x & (kSubRows - 1)) == 0) sgr[(size_t)blockIdx.x * kSub + (threadIdx.x / kSubRows)] = glo | (ghi << 16);
    glo = min(glo, shfl_xor_u(glo, 32)); ghi = max(ghi, shfl_xor_u(ghi, 32));
    if ((threadIdx.x & 63) == 0) gred[threadIdx.x >> 6] = make_uint2(glo, ghi);
  }
  if ((threadIdx.x & (kSubRows - 1)) == 0)
    sbox[(size_t)blockIdx.x * kSub + (threadIdx.x / kSubRows)] = make_uint2(xl | (yl << 16), (xh | (yh << 16)) | kGuard);
  static_assert(kSubRows == 32, "one more exchange joins the two sub-tiles of a wave");
  xl = min(xl, shfl_xor_u(xl, 32)); yl = min(yl, shfl_xor_u(yl, 32));
  xh = max(xh, shfl_xor_u(xh, 32)); yh = max(yh, shfl_xor_u(yh, 32));
  if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = make_uint4(xl, yl, xh, yh);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint4 m = wred[0];
    for (int k = 1; k < 4; ++k) {
      m.x = min(m.x, wred[k].x); m.y = min(m.y, wred[k].y);
      m.z = max(m.z, wred[k].z); m.w = max(m.w, wred[k].w);
    }
    // a tile of sentinels only keeps lo = max, hi = 0: it meets nothing
    tbox[blockIdx.x] = make_uint2(m.x | (m.y << 16), (m.z | (m.w << 16)) | kGuard);
    // bounding box (and bucket range) of the SUPERTILE of kSuper consecutive tiles: every field is kept
    // as a maximum (the lows complemented) so that the zero-filled words mean "empty"
    unsigned int *sp = super ? super + (size_t)(blockIdx.x / kSuper) * 8 : nullptr;
    if (sp) {
      atomicMax(sp + 0, 0x7FFFu - m.x); atomicMax(sp + 1, 0x7FFFu - m.y);
      atomicMax(sp + 2, m.z); atomicMax(sp + 3, m.w);
    }
    if (pg) {
      uint2 r = gred[0];
      for (int k = 1; k < 4; ++k) { r.x = min(r.x, gred[k].x); r.y = max(r.y, gred[k].y); }
      tgr[blockIdx.x] = r.x | (r.y << 16);
      if (sp) { atomicMax(sp + 4, 0xFFFFu - r.x); atomicMax(sp + 5, r.y); }
    }
  }
}

// packed interval test: every guard bit of (b.hi|G - a.lo) and (a.hi|G - b.lo)
// survives  <=>  the intervals meet on both axes.  Returns 0 for "meet".
__device__ __forceinline__ uint32_t qdisjoint(uint32_t alo, uint32_t ahiG, uint32_t blo, uint32_t bhiG) {
  const uint32_t d1 = bhiG - alo, d2 = ahiG - blo;
  return (~d1 | ~d2) & kGuard;  // one v_bitop3_b32
}

// ---- 5: unit culling --------------------------------------------------------------------
// A scan UNIT is (row sub-tile = 32 consecutive positions of tile A, column tile B >= A):
// one wave's worth of work in the scan kernel.  Block per row tile A, threads over B;
// a tile pair whose boxes meet emits a unit for every sub-tile of A that meets B's box,
// stored as { A << 16 | B, sub-tile }  [ntiles < 65536].
// The scan kernel's duration is set by its slowest wave, and the work of a unit differs
// by an order of magnitude (a sub-tile against its own tile vs. one that merely touches
// the column tile) -- so units that overlap substantially ("heavy") are listed from the
// FRONT of the list and start first, the others from the BACK (longest-processing-
// time-first scheduling).
__device__ __forceinline__ float qbox_area(uint32_t lo, uint32_t hiG) {
  const uint32_t w = hiG - lo;   // per-axis width, guard bits survive (hi >= lo)
  return (float)((w & 0x7FFFu) + 1u) * (float)(((w >> 16) & 0x7FFFu) + 1u);
}

__device__ __forceinline__ bool ranges_meet(uint32_t a, uint32_t b) {   // lo | hi << 16 each
  return (a & 0xFFFFu) <= (b >> 16) && (b & 0xFFFFu) <= (a >> 16);
}

__global__ __launch_bounds__(256) void nms_cull_kernel(const uint2 *__restrict__ tbox,
                                                       const uint2 *__restrict__ sbox,
                                                       const uint32_t *__restrict__ tgr,
                                                       const uint32_t *__restrict__ sgr, int nt,
                                                       uint2 *__restrict__ units,
                                                       unsigned int scap,
                                                       unsigned int *__restrict__ ucnts,
                                                       const unsigned int *__restrict__ super) {
  // survivors are collected in LDS and published with two global atomics per block, on
  // the counters of this tile's shard of the unit list
  __shared__ uint32_t lists[kCullChunk * kSub];   // sub << 16 | B; heavy from the front, light from the back
  __shared__ unsigned int nh, nl, gbase_h, gbase_l, nsup;
  __shared__ uint16_t slist[kCullChunk / kSuper + 2];
  const int A = blockIdx.x;
  const int b0 = A + (int)blockIdx.y * kCullChunk;   // this block's column tiles [b0, b1)
  if (b0 >= nt) return;
  const int b1 = min(nt, b0 + kCullChunk);
  const int span = (b1 - b0) * kSub;
  if (threadIdx.x == 0) { nh = 0; nl = 0; nsup = 0; }
  __syncthreads();
  const uint2 a = tbox[A];
  // first the supertiles of the chunk (<= 65): a spatially sorted scene meets a handful of them
  // (small calls have no supertiles -- one more barrier and a round of atomics cost 1.7 us at
  // 100k boxes -- and visit every tile of the chunk)
  if (super) {
    const int s0 = b0 / kSuper, ns = (b1 - 1) / kSuper - s0 + 1;
    const uint32_t gaT = tgr ? tgr[A] : 0xFFFF0000u;
    if ((int)threadIdx.x < ns) {
      const uint4 *sp = reinterpret_cast<const uint4 *>(super + (size_t)(s0 + threadIdx.x) * 8);
      const uint4 bx = sp[0];
      const uint32_t lo = (0x7FFFu - bx.x) | ((0x7FFFu - bx.y) << 16), hi = (bx.z | (bx.w << 16)) | kGuard;
      bool meet = bx.x + bx.z >= 0x7FFFu && qdisjoint(a.x, a.y, lo, hi) == 0u;   // (lo <= hi: not empty)
      if (meet && tgr) {
        const uint4 gx = sp[1];
        meet = ranges_meet(gaT, (0xFFFFu - gx.x) | (gx.y << 16));
      }
      if (meet) slist[atomicAdd(&nsup, 1u)] = (uint16_t)(s0 + threadIdx.x);
    }
    __syncthreads();
  }
  const int ncheck = super ? (int)nsup * kSuper : b1 - b0;
  uint2 sa[kSub];
#pragma unroll
  for (int w = 0; w < kSub; ++w) sa[w] = sbox[(size_t)A * kSub + w];
  uint32_t ga[kSub];   // grouped calls: bucket range of every sub-tile (else "everything")
#pragma unroll
  for (int w = 0; w < kSub; ++w) ga[w] = sgr ? sgr[(size_t)A * kSub + w] : 0xFFFF0000u;
  for (int it = threadIdx.x; it < ncheck; it += 256) {
    const int B = super ? (int)slist[it / kSuper] * kSuper + (it % kSuper) : b0 + it;
    if (B < b0 || B >= b1 || B == A) continue;                       // (diagonal units are implicit)
    const uint2 b = tbox[B];
    if (qdisjoint(a.x, a.y, b.x, b.y) != 0u) continue;
    const uint32_t gb = tgr ? tgr[B] : 0xFFFF0000u;
#pragma unroll
    for (int w = 0; w < kSub; ++w) {
      if (qdisjoint(sa[w].x, sa[w].y, b.x, b.y) != 0u || !ranges_meet(ga[w], gb)) continue;
      // shared part of the two boxes vs. the sub-tile's own box: most rows active?
      const uint32_t ilo = __builtin_elementwise_max(sa[w].x & 0x7FFFu, b.x & 0x7FFFu) |
                           (__builtin_elementwise_max(sa[w].x >> 16, b.x >> 16) << 16);
      const uint32_t ihi = __builtin_elementwise_min(sa[w].y & 0x7FFFu, b.y & 0x7FFFu) |
                           (__builtin_elementwise_min((sa[w].y >> 16) & 0x7FFFu, (b.y >> 16) & 0x7FFFu) << 16) | kGuard;
      const bool heavy = qbox_area(ilo, ihi) * 4.0f >= qbox_area(sa[w].x, sa[w].y);
      const uint32_t v = ((uint32_t)w << 16) | (uint32_t)B;
      if (heavy) lists[atomicAdd(&nh, 1u)] = v;
      else lists[span - 1 - atomicAdd(&nl, 1u)] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int sh = (unsigned int)A & (kShards - 1);
    gbase_h = nh ? atomicAdd(&ucnts[sh * kUnitStride], nh) : 0u;
    gbase_l = nl ? atomicAdd(&ucnts[(kShards + sh) * kUnitStride], nl) : 0u;
  }
  __syncthreads();
  // a segment that does not fit is reported to the host (heavy + light > scap, see
  // nms_compact_kernel); never write outside it
  uint2 *seg = units + (size_t)((unsigned int)A & (kShards - 1)) * scap;
  for (unsigned int k = threadIdx.x; k < nh; k += 256) {
    const uint32_t v = lists[k];
    if (gbase_h + k < scap) seg[gbase_h + k] = make_uint2(((uint32_t)A << 16) | (v & 0xFFFFu), v >> 16);
  }
  for (unsigned int k = threadIdx.x; k < nl; k += 256) {
    const uint32_t v = lists[span - 1 - k];
    if (gbase_l + k < scap) seg[scap - 1u - (gbase_l + k)] = make_uint2(((uint32_t)A << 16) | (v & 0xFFFFu), v >> 16);
  }
}

// ---- 4 + 5 in one launch (calls of fewer than kSuperMinTiles tiles) ---------------------------------------------
// A block takes a TICKET (its rank in start order), reduces the boxes of tile A = ticket out of the sorted
// intervals, publishes the tile box as ONE 8-byte word (its guard bits double as the VALID flag: the words are
// zero-filled by the frame kernel) and lists the units (sub-tile of A, tile B) for all B < A whose boxes meet --
// the mirror image of nms_cull_kernel's B > A, the same set of tile pairs.  Tiles B < A belong to tickets that
// were handed out earlier: those blocks are running and publish without waiting for anyone, so a poll always
// ends (bounded anyway; a timeout is reported as a unit-list overflow).  Visibility: the payload IS the flagged
// word (agent-scope store / load), the bucket range of a grouped call is stored before it behind an
// s_waitcnt vmcnt(0) and read with agent-scope loads after it (MI355X_MICROARCH.md, handoff-flag).
__global__ __launch_bounds__(256) void nms_cullbox_kernel(const uint2 *__restrict__ qs, int64_t n,
                                                          const int32_t *__restrict__ pg,
                                                          const NmsFrame *__restrict__ frame, int nt,
                                                          uint2 *__restrict__ tbox, uint32_t *__restrict__ tgr,
                                                          uint2 *__restrict__ units, unsigned int scap,
                                                          unsigned int *__restrict__ ucnts,
                                                          unsigned int *__restrict__ ticket) {
  __shared__ uint32_t lists[kCullChunk * kSub];   // sub << 16 | B; heavy from the front, light from the back
  __shared__ unsigned int nh, nl, gbase_h, gbase_l, s_ticket;
  __shared__ uint4 wred[4];
  __shared__ uint2 gred[4];
  __shared__ uint2 s_sub[kSub];
  __shared__ uint32_t s_sgr[kSub];
  __shared__ uint2 s_tile;
  __shared__ uint32_t s_tgr;
  const int tid = threadIdx.x;
  if (tid == 0) { s_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); nh = 0; nl = 0; }
  __syncthreads();
  const int A = (int)s_ticket;
  // ---- the boxes of tile A
  const int64_t s = (int64_t)A * kTile + tid;
  const uint2 o = qs[s];
  uint32_t glo = 0xFFFFu, ghi = 0u;
  if (pg && s < n && o.x != kSentLo) { glo = ghi = group_bucket(pg[s], frame->gbits); }
  uint32_t xl = o.x & 0x7FFFu, yl = (o.x >> 16) & 0x7FFFu;
  uint32_t xh = o.y & 0x7FFFu, yh = (o.y >> 16) & 0x7FFFu;
  for (int k = kSubRows / 2; k > 0; k >>= 1) {
    xl = min(xl, shfl_xor_u(xl, k)); yl = min(yl, shfl_xor_u(yl, k));
    xh = max(xh, shfl_xor_u(xh, k)); yh = max(yh, shfl_xor_u(yh, k));
    glo = min(glo, shfl_xor_u(glo, k)); ghi = max(ghi, shfl_xor_u(ghi, k));
  }
  if ((tid & (kSubRows - 1)) == 0) {
    s_sub[tid / kSubRows] = make_uint2(xl | (yl << 16), (xh | (yh << 16)) | kGuard);
    s_sgr[tid / kSubRows] = pg ? (glo | (ghi << 16)) : 0xFFFF0000u;
  }
  xl = min(xl, shfl_xor_u(xl, 32)); yl = min(yl, shfl_xor_u(yl, 32));
  xh = max(xh, shfl_xor_u(xh, 32)); yh = max(yh, shfl_xor_u(yh, 32));
  glo = min(glo, shfl_xor_u(glo, 32)); ghi = max(ghi, shfl_xor_u(ghi, 32));
  if ((tid & 63) == 0) { wred[tid >> 6] = make_uint4(xl, yl, xh, yh); gred[tid >> 6] = make_uint2(glo, ghi); }
  __syncthreads();
  if (tid == 0) {
    uint4 m = wred[0];
    uint2 r = gred[0];
    for (int k = 1; k < 4; ++k) {
      m.x = min(m.x, wred[k].x); m.y = min(m.y, wred[k].y);
      m.z = max(m.z, wred[k].z); m.w = max(m.w, wred[k].w);
      r.x = min(r.x, gred[k].x); r.y = max(r.y, gred[k].y);
    }
    const uint2 tb = make_uint2(m.x | (m.y << 16), (m.z | (m.w << 16)) | kGuard);
    const uint32_t tg = pg ? (r.x | (r.y << 16)) : 0xFFFF0000u;
    s_tile = tb; s_tgr = tg;
    if (pg) {
      __hip_atomic_store(&tgr[A], tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&tbox[A]),
                       (unsigned long long)tb.x | ((unsigned long long)tb.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- units (sub-tile of A, tile B) for B < A
  const uint2 a = s_tile;
  const uint32_t gaT = s_tgr;
  (void)gaT;
  uint2 sa[kSub];
  uint32_t ga[kSub];
#pragma unroll
  for (int w = 0; w < kSub; ++w) { sa[w] = s_sub[w]; ga[w] = s_sgr[w]; }
  const int span = A * kSub;   // (A < kCullChunk: the fused kernel serves calls of < kSuperMinTiles tiles)
  bool timeout = false;
  for (int B = tid; B < A; B += 256) {
    unsigned long long raw;
    int spins = 0;
    do {
      raw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&tbox[B]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while (!((raw >> 32) & 0x80000000ull) && ++spins < (1 << 22));
    if (!((raw >> 32) & 0x80000000ull)) { timeout = true; continue; }
    const uint2 b = make_uint2((uint32_t)raw, (uint32_t)(raw >> 32));
    if (qdisjoint(a.x, a.y, b.x, b.y) != 0u) continue;
    const uint32_t gb = pg ? __hip_atomic_load(&tgr[B], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFF0000u;
#pragma unroll
    for (int w = 0; w < kSub; ++w) {
      if (qdisjoint(sa[w].x, sa[w].y, b.x, b.y) != 0u || !ranges_meet(ga[w], gb)) continue;
      // shared part of the two boxes vs. the sub-tile's own box: most rows active?
      const uint32_t ilo = __builtin_elementwise_max(sa[w].x & 0x7FFFu, b.x & 0x7FFFu) |
                           (__builtin_elementwise_max(sa[w].x >> 16, b.x >> 16) << 16);
      const uint32_t ihi = __builtin_elementwise_min(sa[w].y & 0x7FFFu, b.y & 0x7FFFu) |
                           (__builtin_elementwise_min((sa[w].y >> 16) & 0x7FFFu, (b.y >> 16) & 0x7FFFu) << 16) | kGuard;
      const bool heavy = qbox_area(ilo, ihi) * 4.0f >= qbox_area(sa[w].x, sa[w].y);
      const uint32_t v = ((uint32_t)w << 16) | (uint32_t)B;
      if (heavy) lists[atomicAdd(&nh, 1u)] = v;
      else lists[span - 1 - atomicAdd(&nl, 1u)] = v;
    }
  }
  const unsigned int sh = (unsigned int)A & (kShards - 1);
  if (timeout) atomicMax(&ucnts[sh * kUnitStride], 0x7FFFFFFFu);   // "cannot happen": reported as an overflow of the unit list
  __syncthreads();
  if (tid == 0) {
    gbase_h = nh ? atomicAdd(&ucnts[sh * kUnitStride], nh) : 0u;
    gbase_l = nl ? atomicAdd(&ucnts[(kShards + sh) * kUnitStride], nl) : 0u;
  }
  __syncthreads();
  // a segment that does not fit is reported to the host (heavy + light > scap, see
  // nms_compact_kernel); never write outside it
  uint2 *seg = units + (size_t)sh * scap;
  for (unsigned int k = tid; k < nh; k += 256) {
    const uint32_t v = lists[k];
    if (gbase_h + k < scap) seg[gbase_h + k] = make_uint2(((uint32_t)A << 16) | (v & 0xFFFFu), v >> 16);
  }
  for (unsigned int k = tid; k < nl; k += 256) {
    const uint32_t v = lists[span - 1 - k];
    if (gbase_l + k < scap) seg[scap - 1u - (gbase_l + k)] = make_uint2(((uint32_t)A << 16) | (v & 0xFFFFu), v >> 16);
  }
}

// ---- 6: scan kernel --------------------------------------------------------------------
// One wave = one unit: 32 rows of tile A (lane l holds row l; broadcast with
// v_readlane) x 256 columns of tile B (4 per lane), and only the rows whose interval
// meets B's bounding box are visited.  15 full-rate VALU + 2 v_readlane per row.
// Rows with a candidate are only RECORDED in `hit` and pushed out of line.
// Threshold-aware pruning of a candidate (never changes the result: a pair is
// dropped only when its overlap provably stays below thr in the fp32 spec):
//   * inter <= min(aA, aB)            =>  iou <= min/max
//   * inter <= area(AABB_A n AABB_B)  =>  iou <= I/(aA + aB - I),  I = that area, bounded
//     from above through the quantised intervals (cells shared + 1) * cell size
// both with a 1e-6 relative margin (the spec's roundings are a few 2^-24).
struct Prune {
  float thr;       // <= 0: pruning off
  float cpx, cpy;  // pixels per quantisation cell (rounded up)
};

// does a packed interval touch cell 0 or the last cell on either axis?
__device__ __forceinline__ bool qedge(uint32_t lo, uint32_t hiG) {
  return (lo & 0x7FFFu) == 0u || (lo & 0x7FFF0000u) == 0u || (hiG & 0x7FFFu) == kQMax || ((hiG >> 16) & 0x7FFFu) == kQMax;
}

// qedge(a) || qedge(b) in nine operations: a 15-bit field that is 0 borrows into its guard position when 1 is taken from
// it (bits 15 / 31 of `lo` are clear), one that is kQMax = 0x7FFE carries out of it when 2 are added (they are set in `hiG`); a
// borrow / carry that runs on into the upper field can only turn it into an edge when the lower field already is one
__device__ __forceinline__ bool qedge2(uint32_t alo, uint32_t ahiG, uint32_t blo, uint32_t bhiG) {
  static_assert(kQMax == 0x7FFEu, "the carry below");
  const uint32_t k = 0x00010001u, k2 = 0x00020002u;
  return ((((alo - k) | (blo - k)) | ~((ahiG + k2) & (bhiG + k2))) & 0x80008000u) != 0u;
}

// first (cheap) bound: iou <= min(area)/max(area)
__device__ __forceinline__ bool prune_area(const Prune &pr, float ra, float ca) {
  const float mn = min_sel(ra, ca), mx = max_sel(ra, ca);
  return !(mn < (pr.thr * mx) * 0.999999f);
}

// second bound: inter <= area of the AABB overlap, bounded through the shared cells
__device__ __forceinline__ bool prune_aabb(const Prune &pr, uint32_t rlo, uint32_t rhiG, float ra,
                                           uint32_t clo, uint32_t chiG, float ca) {
  // shared cells per axis: min(b.hi - a.lo, a.hi - b.lo, own widths) + 1
  const uint32_t d1 = chiG - rlo, d2 = rhiG - clo;          // guard bits set (candidate)
  const uint32_t wr = rhiG - rlo, wc = chiG - clo;
  const int sx = (int)min(min(d1 & 0x7FFFu, d2 & 0x7FFFu), min(wr & 0x7FFFu, wc & 0x7FFFu)) + 1;
  const int sy = (int)min(min((d1 >> 16) & 0x7FFFu, (d2 >> 16) & 0x7FFFu),
                          min((wr >> 16) & 0x7FFFu, (wc >> 16) & 0x7FFFu)) + 1;
  const float I = ((float)sx * pr.cpx) * ((float)sy * pr.cpy) * 1.000001f;   // >= true AABB overlap
  // possible hit only if I / (ra + ca - I) >= thr  <=>  I (1 + thr) >= thr (ra + ca)
  return I * (1.0f + pr.thr) >= (pr.thr * (ra + ca)) * 0.999999f;
}

// ---- narrow-phase helpers (clip kernel, the direct scan and the in-line clip of the sorted scan) ------------------------------
__device__ __forceinline__ void load_rec(const QuadRec *__restrict__ rec, uint32_t p, Quad &q,
                                         uint32_t &key, int &grp, float &thr) {
  const float4 *r = reinterpret_cast<const float4 *>(rec + p);
  float4 a = r[0], b = r[1], c = r[2], d = r[3];
  q.ox = a.x; q.oy = a.y; q.ex = a.z; q.ey = a.w;
  q.rx[0] = b.x; q.rx[1] = b.y; q.rx[2] = b.z; q.rx[3] = b.w;
  q.ry[0] = c.x; q.ry[1] = c.y; q.ry[2] = c.z; q.ry[3] = c.w;
  q.area = d.x;
  key = __float_as_uint(d.y);
  grp = __float_as_int(d.z);
  thr = d.w;
}

__device__ __forceinline__ void swap_if(bool c, float &a, float &b) {
  const float t = c ? b : a;
  b = c ? a : b;
  a = t;
}

// One candidate pair: the edge points from the box that PRECEDES in the greedy order
// (smaller score key, then smaller index) to the other one; the clip always takes the
// preceding box as the subject A, exactly like the i-before-j loop of nms_cpu.cpp:35-58.
__device__ __forceinline__ bool narrow_hit(const Quad &A, const Quad &B, float thr, int flags) {
  float ov;
  bool rej = false;
  if (flags & kFlagF64) {       // (wave-uniform) nms_cpu_kernel<double>: `ovr >= threshold` with the float threshold promoted
    const double ovd = hbb_overlap_f64(A, B), td = (double)thr;
    return (flags & kFlagStrict) ? (ovd > td) : (ovd >= td);
  }
  if (flags & kFlagHbb) ov = hbb_overlap(A.rx, A.area, B.rx, B.area);
  else ov = quad_overlap(A, B, 0, rej);
  const bool over = (flags & kFlagStrict) ? (ov > thr) : (ov >= thr);
  return over && !rej;
}

__device__ __forceinline__ bool clip_loaded(Quad &A, Quad &B, uint32_t ka, uint32_t kb, int ga, int gb,
                                            float ta, float tb, uint2 pr, int flags, uint32_t &src, uint32_t &dst,
                                            const uint32_t *__restrict__ klo = nullptr) {
  bool b_first = (kb < ka) || (kb == ka && pr.y < pr.x);
  if ((flags & kFlagF64) && kb == ka) {      // equal high words: the low words of the 64-bit score keys decide
    const uint32_t la = klo[pr.x], lb = klo[pr.y];
    b_first = (lb < la) || (lb == la && pr.y < pr.x);
  }
  const float thr = b_first ? tb : ta;  // the earlier box's threshold
  src = b_first ? pr.y : pr.x;
  dst = b_first ? pr.x : pr.y;
  swap_if(b_first, A.ox, B.ox); swap_if(b_first, A.oy, B.oy);
  swap_if(b_first, A.ex, B.ex); swap_if(b_first, A.ey, B.ey);
  swap_if(b_first, A.area, B.area);
#pragma unroll
  for (int k = 0; k < 4; ++k) { swap_if(b_first, A.rx[k], B.rx[k]); swap_if(b_first, A.ry[k], B.ry[k]); }
  return (ga == gb) && narrow_hit(A, B, thr, flags);
}

__device__ __forceinline__ bool clip_pair(const QuadRec *__restrict__ rec, uint2 pr, int flags, uint32_t &src,
                                          uint32_t &dst, const uint32_t *__restrict__ klo = nullptr) {
  Quad A, B;
  uint32_t ka, kb;
  int ga, gb;
  float ta, tb;
  load_rec(rec, pr.x, A, ka, ga, ta);
  load_rec(rec, pr.y, B, kb, gb, tb);
  return clip_loaded(A, B, ka, kb, ga, gb, ta, tb, pr, flags, src, dst, klo);
}

// hits of one wave -> its shard's edge segment (+ the in-degree of the targets)
__device__ __forceinline__ void push_edges(bool hit, uint32_t src, uint32_t dst, uint2 *__restrict__ eseg,
                                           unsigned long long segcap,
                                           unsigned long long *__restrict__ ecnt, int *__restrict__ cnt,
                                           int lane) {
  const uint64_t mk = __ballot(hit);
  if (mk) {
    const int leader = __ffsll((long long)mk) - 1;
    unsigned long long pos0 = 0;
    if (lane == leader) pos0 = atomicAdd(ecnt, (unsigned long long)__popcll(mk));
    pos0 = __shfl(pos0, leader);
    if (hit) {
      const unsigned long long pos = pos0 + __popcll(mk & ((1ull << lane) - 1ull));
      if (pos < segcap) eseg[pos] = make_uint2(src, dst);
      atomicAdd(&cnt[dst], 1);
    }
  }
}

// The per-wave LDS queue holds RAW candidates as spatial positions (row, column).  A
// batch of 64 is pruned with one candidate per lane (the two bounds cost ~40 VALU: per
// batch, not per candidate), and the survivors go to the wave's global candidate list
// as (box id, box id) with one returning atomic per batch.
//
// INLINE (n up to a few 10^5, where the call is a chain of latency-bound launches): the
// survivors are not listed but collected in a second per-wave LDS queue and CLIPPED by this
// wave as soon as 64 of them are there (full lanes in dense scenes; one partial batch at the
// end of the wave in sparse ones) -- the clip launch, its candidate list round trip and its
// load chain disappear (100k sparse boxes: scan 21 + clip 17 + 2.5 us gap -> one kernel).
constexpr int kClipQueue = 128;   // carry (<= 63) + one pruned batch (<= 64)

template <bool INLINE>
struct SortedSink {
  Prune pr;
  const uint4 *__restrict__ pf;             // per position: packed interval, area, box id
  const int32_t *__restrict__ pg;           // per position: group id (grouped calls, else null)
  const float *__restrict__ pt;             // per position: threshold (per-box thresholds, else null)
  uint2 *__restrict__ cand;                 // this wave's candidate segment
  unsigned long long segcap;
  unsigned long long *__restrict__ ccnt;
  // INLINE only
  const QuadRec *__restrict__ rec;
  const uint32_t *__restrict__ klo;         // low words of 64-bit score keys (kFlagF64, else null)
  int flags;
  uint2 *__restrict__ eseg;                 // this wave's edge segment
  unsigned long long *__restrict__ ecnt;
  int *__restrict__ cnt;
  uint2 *cq;                                // LDS: pruned candidates waiting for a full clip batch
  int cqn;

  __device__ __forceinline__ void clip_batch(const uint2 *b, int count, int lane) const {
    bool hit = false;
    uint32_t src = 0, dst = 0;
    if (lane < count) hit = clip_pair(rec, b[lane], flags, src, dst, klo);
    push_edges(hit, src, dst, eseg, segcap, ecnt, cnt, lane);
  }
  __device__ __forceinline__ void finish(int lane) {   // end of the wave: the partial batch
    if (INLINE && cqn > 0) {
      __builtin_amdgcn_wave_barrier();
      clip_batch(cq, cqn, lane);
      cqn = 0;
    }
  }

  __device__ __forceinline__ void flush(const uint2 *q, int count, int lane) {
    bool keep = lane < count;
    uint2 ids = make_uint2(0u, 0u);
    if (keep) {
      const uint2 e = q[lane];
      const uint4 a = pf[e.x], b = pf[e.y];
      const float ra = __uint_as_float(a.z), ca = __uint_as_float(b.z);
      ids = make_uint2(a.w, b.w);
      if (pg) keep = pg[e.x] == pg[e.y];   // boxes of different groups never interact
      // per-box thresholds: the one that applies is the earlier box's, i.e. one of the two --
      // a pair that cannot reach the SMALLER one cannot reach it either
      Prune p = pr;
      if (pt) p.thr = min_sel(pt[e.x], pt[e.y]);
      if (keep && p.thr > 0.0f) {
        // an interval that touches the first or last cell may have been CLAMPED into the frame
        // (robust frame, nms_quant_kernel): its cells then say nothing about its extent
        const bool edge = qedge(a.x, a.y) || qedge(b.x, b.y);
        keep = prune_area(p, ra, ca) && (edge || prune_aabb(p, a.x, a.y, ra, b.x, b.y, ca));
      }
    }
    const uint64_t mk = __ballot(keep);
    if (mk == 0) return;
    if (INLINE) {
      if (lane == 0) atomicAdd(ccnt, (unsigned long long)__popcll(mk));   // statistics only
      if (keep) cq[cqn + __popcll(mk & ((1ull << lane) - 1ull))] = ids;
      cqn += __popcll(mk);
      if (cqn >= 64) {
        cqn -= 64;
        __builtin_amdgcn_wave_barrier();
        clip_batch(cq + cqn, 64, lane);
      }
      return;
    }
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(ccnt, (unsigned long long)__popcll(mk));
    base = __shfl(base, 0) + (unsigned long long)__popcll(mk & ((1ull << lane) - 1ull));
    if (keep && base < segcap) cand[base] = ids;
  }
};

template <bool DIAG, class Sink>
__device__ __forceinline__ void nms_scan_rows(uint2 rv, uint32_t rows, uint32_t sa0, uint32_t sb0,
                                              const uint2 (&col)[4], Sink &sink, uint2 *q, int &qn,
                                              int lane) {
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t sbl = sb0 + lane * kNmsColsPerLane;
  while (rows) {   // `rows`: the rows of this sub-tile that meet the column tile's box
    // a chunk of rows: the per-pair results stay in registers, the lanes with a candidate
    // come out of the vote as a scalar mask
    uint32_t t[kRowChunk][4];
    uint64_t lm[kRowChunk];
    int rr[kRowChunk];
#pragma unroll
    for (int u = 0; u < kRowChunk; ++u) {
      // lane kSubRows holds a sentinel row (never a candidate): padding of the last chunk
      const int r = rows ? __builtin_ctz(rows) : kSubRows;
      rows &= rows - 1u;
      rr[u] = r;
      const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane((int)rv.x, r);
      const uint32_t rhi = (uint32_t)__builtin_amdgcn_readlane((int)rv.y, r);
#pragma unroll
      for (int c = 0; c < 4; ++c) t[u][c] = qdisjoint(rlo, rhi, col[c].x, col[c].y);
      if (DIAG) {  // only pairs (row, col) with col > row exist in a diagonal tile
        const uint32_t sa = sa0 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) t[u][c] |= (sbl + c > sa) ? 0u : 1u;
      }
      const uint32_t m = min(min(t[u][0], t[u][1]), min(t[u][2], t[u][3]));
      lm[u] = __ballot(m == 0u);
    }
#pragma unroll
    for (int u = 0; u < kRowChunk; ++u) {
      if (lm[u] == 0) continue;   // (wave-uniform) no candidate in this row: the usual case
      const uint32_t sa = sa0 + (uint32_t)rr[u];
      uint64_t mk = lm[u];
      if (__popcll(mk) <= 2) {
        // one or two lanes: read their four results back and push from scalar registers
        while (mk) {
          const int l = __builtin_ctzll(mk);
          mk &= mk - 1ull;
          const uint32_t sb = sb0 + (uint32_t)l * kNmsColsPerLane;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (__builtin_amdgcn_readlane((int)t[u][c], l) == 0) {
              if (lane == 0) q[qn] = make_uint2(sa, sb + (uint32_t)c);
              ++qn;
            }
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < kNmsColsPerLane; ++c) {
          const bool cnd = t[u][c] == 0u;
          const uint64_t mc = __ballot(cnd);
          if (cnd) q[qn + __popcll(mc & lt_mask)] = make_uint2(sa, sbl + c);
          qn += __popcll(mc);
        }
      }
      while (qn >= 64) {  // prune + flush a full batch
        qn -= 64;
        __builtin_amdgcn_wave_barrier();
        sink.flush(q + qn, 64, lane);
      }
    }
  }
}

// Wave g handles units g, g + (waves in the grid), ... of
//   [ diagonal units (sub-tile vs. its own tile, implicit) | heavy ... | light ... ].
// Consecutive blocks land on different CUs, so this deals the expensive units evenly.
template <bool INLINE>
__global__ __launch_bounds__(64 * kNmsWaves) void nms_scan_kernel(const uint2 *__restrict__ qs,
                                                       const uint4 *__restrict__ pf,
                                                       const int32_t *__restrict__ pg,
                                                       const float *__restrict__ pt,
                                                       const float *__restrict__ cellpx, float thr,
                                                       const uint2 *__restrict__ tbox, int nt,
                                                       const uint2 *__restrict__ units,
                                                       unsigned int scap,
                                                       const unsigned int *__restrict__ ucnts,
                                                       uint2 *__restrict__ cands,
                                                       unsigned long long segcap,
                                                       unsigned long long *__restrict__ ccnts,
                                                       const QuadRec *__restrict__ rec, int flags,
                                                       uint2 *__restrict__ edges,
                                                       int *__restrict__ cnt, const uint32_t *__restrict__ klo) {
  __shared__ uint2 queue[kNmsWaves][kNmsQueue];
  __shared__ uint2 clipq[INLINE ? kNmsWaves : 1][INLINE ? kClipQueue : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint2 *q = queue[wave];
  int qn = 0;
  const uint32_t gw = blockIdx.x * kNmsWaves + wave;       // global wave id
  const uint32_t nw = gridDim.x * kNmsWaves;
  const int shard = (int)(gw & (kShards - 1));
  SortedSink<INLINE> sink;
  sink.pr.thr = thr; sink.pr.cpx = cellpx[0]; sink.pr.cpy = cellpx[1];
  sink.pf = pf; sink.pg = pg; sink.pt = pt;
  sink.cand = cands + (size_t)shard * segcap;
  sink.segcap = segcap;
  // (INLINE: the candidates are only counted, in the statistics slot -- their list does not exist)
  sink.ccnt = ccnts + (size_t)((INLINE ? 2 * kShards : 0) + shard) * kCandStride;
  sink.rec = rec; sink.flags = flags; sink.klo = klo;
  sink.eseg = edges + (size_t)shard * segcap;
  sink.ecnt = ccnts + (size_t)(kShards + shard) * kCandStride;
  sink.cnt = cnt;
  sink.cq = clipq[INLINE ? wave : 0];
  sink.cqn = 0;
  // lane s holds shard s of the unit list: its counts and their exclusive prefix sums
  const unsigned int cnt_h = ucnts[lane * kUnitStride], cnt_l = ucnts[(kShards + lane) * kUnitStride];
  if (__ballot(cnt_h + cnt_l > scap) != 0ull) return;   // unit list overflow: reported by the host, nothing valid to scan
  unsigned int inc_h = cnt_h, inc_l = cnt_l;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int th = __shfl_up(inc_h, o), tl = __shfl_up(inc_l, o);
    if (lane >= o) { inc_h += th; inc_l += tl; }
  }
  const unsigned int ex_h = inc_h - cnt_h, ex_l = inc_l - cnt_l;
  const unsigned int nheavy = __builtin_amdgcn_readlane(inc_h, 63), nlight = __builtin_amdgcn_readlane(inc_l, 63);
  const unsigned int ndiag = (unsigned int)nt * kSub;
  const unsigned int nun = ndiag + nheavy + nlight;
  // (Round 3: the loop as a two-stage pipeline -- rows / columns of unit i + 1 and the record of unit i + 2 requested
  // while unit i is scanned, all loads unconditional -- was built and measured: 74 VGPRs (6 waves per SIMD instead of 8),
  // scan phase 24 -> 26-28 us at 100k sparse boxes, 60 -> 72 on the dense scene, 10^6 boxes unchanged.  Not kept: the
  // waves' waiting is spread over the queue flushes and the unit decode, not concentrated in these two loads.)
  for (unsigned int k = gw; k < nun; k += nw) {
    uint32_t A, B, sub;
    if (k < ndiag) {
      A = B = k / kSub; sub = k % kSub;
    } else {
      // all heavy units first (LPT order), then the light ones; the shard of a flat index is
      // the last lane whose exclusive prefix is <= it (empty shards share a prefix with the next)
      const unsigned int j = k - ndiag;
      const bool hv = j < nheavy;
      const unsigned int jj = hv ? j : j - nheavy;
      const int sh = __popcll(__ballot((hv ? ex_h : ex_l) <= jj)) - 1;
      const unsigned int within = jj - (unsigned int)__builtin_amdgcn_readlane(hv ? ex_h : ex_l, sh);
      const uint2 *seg = units + (size_t)sh * scap;
      const uint2 un = hv ? seg[within] : seg[scap - 1u - within];
      A = un.x >> 16; B = un.x & 0xFFFFu; sub = un.y;
    }
    const uint32_t sa0 = A * kTile + sub * kSubRows;         // this unit's 32 rows
    const uint32_t sb0 = B * kTile;                          // 256 columns
    // qs is padded to whole tiles (sentinel intervals never match); lanes >= 32 hold a
    // sentinel row
    const bool rl = lane < kSubRows;
    uint2 rv = make_uint2(kSentLo, kSentHi);
    if (rl) rv = qs[sa0 + lane];
    const uint2 tb = tbox[B];
    uint2 col[4];
    {
      const uint4 *cp = reinterpret_cast<const uint4 *>(qs + sb0 + lane * kNmsColsPerLane);
      const uint4 a = cp[0], b = cp[1];
      col[0] = make_uint2(a.x, a.y); col[1] = make_uint2(a.z, a.w);
      col[2] = make_uint2(b.x, b.y); col[3] = make_uint2(b.z, b.w);
    }
    // only rows whose interval meets the column tile's box can have a candidate
    const uint32_t rows = (uint32_t)__ballot(rl && qdisjoint(rv.x, rv.y, tb.x, tb.y) == 0u);
    if (A == B) nms_scan_rows<true>(rv, rows, sa0, sb0, col, sink, q, qn, lane);
    else nms_scan_rows<false>(rv, rows, sa0, sb0, col, sink, q, qn, lane);
  }
  if (qn > 0) {  // final partial batch, once per wave
    __builtin_amdgcn_wave_barrier();
    sink.flush(q, qn, lane);
  }
  sink.finish(lane);
}

// ---- 6b: direct scan + clip (medium n, no spatial sort) --------------------------------
// Below ~10^4 boxes the spatial sort costs more LAUNCHES (quantise, cell scan, scatter,
// permute, cull: ~5 us each, latency-bound) than it saves work: all nt (nt + 1) / 2 tile
// pairs are scanned.  One wave = up to 32 rows x 256 columns (positions are box ids);
// it quantises its columns in the frame of THEIR bounding box (rows clamp into it --
// floor of a clamped monotone map is monotone, so no real overlap is lost), skips the
// rows that miss that box, and runs the same 3-op packed test as the sorted scan.  A
// batch of 64 raw candidates is pruned on the float intervals AND clipped by the wave
// that found it (the GPU is far from full at these sizes; a separate clip launch would
// cost more than the idle lanes).  grid = (row parts, column tiles).
__device__ __forceinline__ bool prune_aabb_f(const Prune &pr, float4 a, float ra, float4 b, float ca) {
  const float w = max_sel(min_sel(a.y, b.y) - max_sel(a.x, b.x), 0.0f);
  const float h = max_sel(min_sel(a.w, b.w) - max_sel(a.z, b.z), 0.0f);
  const float I = (w * h) * 1.00001f;   // >= true AABB overlap (the intervals carry a margin)
  return I * (1.0f + pr.thr) >= (pr.thr * (ra + ca)) * 0.999999f;
}

// Instrumented build only (-DOBB_GPROBE_STAMPS, tools/exp_gprobe_stamps.py): shader-clock stamps of every wave of the
// grid route's probe kernel, read back with obb_debug_gprobe_stamps (not part of the ABI).
#ifdef OBB_GPROBE_STAMPS
__device__ unsigned long long g_gprobe_stamps[32768 * 10];
#define GP_STAMP(i) do { const uint32_t gw_ = blockIdx.x * kNmsWaves + (threadIdx.x >> 6); if ((threadIdx.x & 63) == 0 && gw_ < 32768u) { \
  g_gprobe_stamps[gw_ * 10 + (i)] = ((i) >= 8) ? wall_clock64() : __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define GP_STAMP(i) do { } while (0)
#endif

struct DirectSink {
  Prune pr;
  bool per_box_thr;                         // thresholds come from the records (pr.thr unused)
  int flags;                                // kFlagStrict | kFlagHbb
  const float4 *__restrict__ iv;
  const float *__restrict__ ar;
  const QuadRec *__restrict__ rec;
  const uint32_t *__restrict__ klo;         // low words of 64-bit score keys (kFlagF64, else null)
  uint2 *__restrict__ eseg;                 // this wave's edge segment
  unsigned long long segcap;
  unsigned long long *__restrict__ ccnt;    // candidate statistics (not an index)
  unsigned long long *__restrict__ ecnt;
  int *__restrict__ cnt;

  __device__ __forceinline__ void flush(const uint2 *q, int count, int lane) const {
    // everything comes from the two 64-byte records (ONE round of loads): the float
    // intervals are rebuilt exactly as the prep kernel built them
    bool keep = lane < count;
    uint2 e = make_uint2(0u, 0u);
    Quad A, B;
    uint32_t ka = 0, kb = 0;
    int ga = 0, gb = 0;
    float ta = 0.0f, tb = 0.0f;
    if (keep) {
      e = q[lane];
      load_rec(rec, e.x, A, ka, ga, ta);
      load_rec(rec, e.y, B, kb, gb, tb);
      float4 a, b;
      nms_interval(A.ox, A.ex, a.x, a.y); nms_interval(A.oy, A.ey, a.z, a.w);
      nms_interval(B.ox, B.ex, b.x, b.y); nms_interval(B.oy, B.ey, b.z, b.w);
      const float big = 3.0e38f;   // (non-finite boxes are isolated, as in nms_prep_kernel)
      const bool fin = (fabs_sel(a.x) < big) && (fabs_sel(a.y) < big) && (fabs_sel(a.z) < big) && (fabs_sel(a.w) < big) &&
                       (fabs_sel(b.x) < big) && (fabs_sel(b.y) < big) && (fabs_sel(b.z) < big) && (fabs_sel(b.w) < big);
      // (the quantised test is coarser than the float intervals: re-test them; boxes of
      // different groups never interact)
      keep = fin && (ga == gb) && (a.x <= b.y) && (b.x <= a.y) && (a.z <= b.w) && (b.z <= a.w);
      Prune p = pr;
      if (per_box_thr) p.thr = min_sel(ta, tb);   // the earlier box's applies: one of the two
      if (keep && p.thr > 0.0f) keep = prune_area(p, A.area, B.area) && prune_aabb_f(p, a, A.area, b, B.area);
    }
    const uint64_t mk = __ballot(keep);
    GP_STAMP(4);
    if (mk == 0) return;
    if (lane == 0) atomicAdd(ccnt, (unsigned long long)__popcll(mk));
    bool hit = false;
    uint32_t src = 0, dst = 0;
    if (keep) hit = clip_loaded(A, B, ka, kb, ga, gb, ta, tb, e, flags, src, dst, klo);
    GP_STAMP(5);
    push_edges(hit, src, dst, eseg, segcap, ecnt, cnt, lane);
    GP_STAMP(6);
  }
};

__device__ __forceinline__ float4 direct_interval(const float4 *__restrict__ iv, uint32_t i, uint32_t n) {
  const float inf = __builtin_huge_valf();
  float4 v = make_float4(inf, -inf, inf, -inf);   // meets nothing
  if (i < n) {
    const float4 t = iv[i];
    if (t.x <= t.y) v = t;   // isolated (non-finite) boxes keep the sentinel
  }
  return v;
}

__device__ __forceinline__ uint32_t qclamp(float x, float b, float s) {
  float u = floorf((x - b) * s);
  u = u < 0.0f ? 0.0f : (u > (float)kQMax ? (float)kQMax : u);
  return (uint32_t)u;
}

__device__ __forceinline__ uint2 direct_quant(float4 v, float bx, float by, float sx, float sy) {
  if (!(v.x <= v.y)) return make_uint2(kSentLo, kSentHi);
  return make_uint2(qclamp(v.x, bx, sx) | (qclamp(v.z, by, sy) << 16),
                    (qclamp(v.y, bx, sx) | (qclamp(v.w, by, sy) << 16)) | kGuard);
}

// rows_per_wave (a power of two <= 32): fewer rows per wave = more waves, for the small
// n where a few hundred 32-row waves would leave most of the chip idle
__global__ __launch_bounds__(64 * kNmsWaves) void nms_scan_direct_kernel(const float4 *__restrict__ iv,
                                                                const float *__restrict__ ar,
                                                                const QuadRec *__restrict__ rec,
                                                                uint32_t n, int rows_per_wave, float thr,
                                                                int per_box_thr, int flags,
                                                                uint2 *__restrict__ edges,
                                                                unsigned long long segcap,
                                                                unsigned long long *__restrict__ ccnts,
                                                                int *__restrict__ cnt, const uint32_t *__restrict__ klo) {
  __shared__ uint2 queue[kNmsWaves][kNmsQueue];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t B = blockIdx.y;
  const uint32_t part = blockIdx.x * kNmsWaves + wave;     // rows [part * rpw, + rpw) of the whole array
  const uint32_t sa0 = part * (uint32_t)rows_per_wave, sb0 = B * kTile;
  const uint32_t A = sa0 / kTile;
  if (A > B || sa0 >= n) return;   // (no block-wide barrier in this kernel)
  uint2 *q = queue[wave];
  int qn = 0;
  const uint32_t gw = (blockIdx.y * gridDim.x + blockIdx.x) * kNmsWaves + wave;
  const int shard = (int)(gw & (kShards - 1));
  DirectSink sink;
  sink.pr.thr = thr; sink.pr.cpx = 0.0f; sink.pr.cpy = 0.0f;
  sink.per_box_thr = per_box_thr != 0;
  sink.flags = flags;
  sink.iv = iv; sink.ar = ar; sink.rec = rec; sink.klo = klo;
  sink.eseg = edges + (size_t)shard * segcap;
  sink.segcap = segcap;
  sink.ccnt = ccnts + (size_t)shard * kCandStride;
  sink.ecnt = ccnts + (size_t)(kShards + shard) * kCandStride;
  sink.cnt = cnt;
  const float4 rvf = direct_interval(iv, (lane < rows_per_wave) ? sa0 + lane : n, n);
  float4 cvf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) cvf[c] = direct_interval(iv, sb0 + lane * kNmsColsPerLane + c, n);
  // frame = bounding box of the columns
  float fxl = min_sel(min_sel(cvf[0].x, cvf[1].x), min_sel(cvf[2].x, cvf[3].x));
  float fxh = max_sel(max_sel(cvf[0].y, cvf[1].y), max_sel(cvf[2].y, cvf[3].y));
  float fyl = min_sel(min_sel(cvf[0].z, cvf[1].z), min_sel(cvf[2].z, cvf[3].z));
  float fyh = max_sel(max_sel(cvf[0].w, cvf[1].w), max_sel(cvf[2].w, cvf[3].w));
  for (int o = 32; o > 0; o >>= 1) {
    fxl = min_sel(fxl, __shfl_xor(fxl, o)); fxh = max_sel(fxh, __shfl_xor(fxh, o));
    fyl = min_sel(fyl, __shfl_xor(fyl, o)); fyh = max_sel(fyh, __shfl_xor(fyh, o));
  }
  if (!(fxl <= fxh)) return;   // no finite column
  const float rx = fxh - fxl, ry = fyh - fyl;
  const float sx = rx > 0.0f ? (float)kQMax / rx : 0.0f, sy = ry > 0.0f ? (float)kQMax / ry : 0.0f;
  uint2 col[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) col[c] = direct_quant(cvf[c], fxl, fyl, sx, sy);
  // rows that meet the columns' box (sentinel rows never do); lanes >= rows_per_wave hold a
  // sentinel, which is also the padding row of nms_scan_rows
  const bool meets = (rvf.x <= fxh) && (fxl <= rvf.y) && (rvf.z <= fyh) && (fyl <= rvf.w);
  const uint2 rv = meets ? direct_quant(rvf, fxl, fyl, sx, sy) : make_uint2(kSentLo, kSentHi);
  const uint32_t rows = (uint32_t)__ballot(meets);
  if (A == B) nms_scan_rows<true>(rv, rows, sa0, sb0, col, sink, q, qn, lane);
  else nms_scan_rows<false>(rv, rows, sa0, sb0, col, sink, q, qn, lane);
  if (qn > 0) {
    __builtin_amdgcn_wave_barrier();
    sink.flush(q, qn, lane);
  }
}

// ---- 6b: grid route, probe + clip -----------------------------------------------------------
// See nms_ginsert_kernel.  A wave owns a CHUNK of 2, 4 or 8 consecutive cells (by the number of cells the frame kernel laid
// out: about 2 048 chunks or more whenever there are that many cells): it stages the chunk's entries in LDS (one
// coalesced 512-byte read per cell), numbers the pairs (s < u) of every cell of the chunk consecutively and gives ONE
// PAIR TO A LANE -- 64 pair tests per step whatever the cells hold.  (The second version gave a cell to a half-wave and
// broadcast entry u to lanes s < u: m - 1 steps for a cell of m entries with m / 64 of the lanes on a pair, 46 % of a
// 30 us kernel at 100k boxes by the stamps of tools/exp_gprobe_stamps.py; the first walked a box's cells from the box's
// own thread: 27 dependent round trips, 42 us.)  A pair is tested where the table entries allow it -- packed intervals
// meet, this cell owns the pair, and for a call with ONE threshold the two bounds of the sorted route's scan (area ratio,
// overlap of the quantised intervals; the entry carries the box's area) -- so that the queue holds pairs that are nearly
// all clipped: DirectSink::flush loads two 64-byte records per pair and runs the clip for a full wave or not at all.
#ifndef OBB_GPROBE_C8
#define OBB_GPROBE_C8 8192
#endif
constexpr int kGridCpwMax = 8;                    // cells per chunk at most
constexpr uint32_t kGridCpw8 = OBB_GPROBE_C8;     // cells laid out from which a chunk is 8 cells
constexpr uint32_t kGridCpw4 = 2048;              // ... 4 cells
constexpr int kGridProbeWaves = 4096;             // most waves of the probe launch (a wave strides over the chunks)

__host__ __device__ inline uint32_t grid_chunk_log2(uint32_t cells) { return cells >= kGridCpw8 ? 3u : (cells >= kGridCpw4 ? 2u : 1u); }

template <bool GROUPED>
__global__ __launch_bounds__(64 * kNmsWaves) void nms_gprobe_kernel(const QuadRec *__restrict__ rec,
                                                           const NmsFrame *__restrict__ frame,
                                                           const uint32_t *__restrict__ gcnt, const uint4 *__restrict__ gtab,
                                                           uint32_t ncells, float thr, int per_box_thr, int flags,
                                                           uint2 *__restrict__ edges, unsigned long long segcap,
                                                           unsigned long long *__restrict__ ccnts, int *__restrict__ cnt,
                                                           const NmsGridCtr *__restrict__ ctr,
                                                           const uint4 *__restrict__ gover, uint32_t over_cap,
                                                           const float *__restrict__ thr_table, int thr_count,
                                                           const float *__restrict__ ar,
                                                           const uint32_t *__restrict__ gbig, uint32_t big_cap, uint32_t n,
                                                           const unsigned long long *__restrict__ hdr_magic, unsigned long long hdr_expect) {
  static_assert(kGridCap == 32, "two cells per 64-lane load");
  if (hdr_magic && *hdr_magic != hdr_expect) return;      // (a retained call whose workspace somebody else wrote: NmsRetain)
  __shared__ uint2 queue[kNmsWaves][kNmsQueue];
  __shared__ uint4 ent[kNmsWaves][kGridCpwMax * kGridCap];
  // pair number t of a cell -> (s, u), s < u, t = u (u - 1) / 2 + s: looked up (the square root and its two corrections were
  // ~15 of a pair step's ~180 instructions), and the chunk's pair offsets for the lanes' cell -> first pair lookup
  constexpr int kPairs = kGridCap * (kGridCap - 1) / 2;
  __shared__ uint16_t pair_su[kPairs];
  __shared__ int pair_first[kNmsWaves][kGridCpwMax];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t gw = blockIdx.x * kNmsWaves + wave, nw = gridDim.x * kNmsWaves;
  GP_STAMP(8);
  GP_STAMP(0);
  for (int t = threadIdx.x; t < kPairs; t += 64 * kNmsWaves) {
    int u = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)t)) * 0.5f);
    int tu = (u * (u - 1)) >> 1;
    if (tu > t) { --u; tu -= u; }
    else if (tu + u <= t) { tu += u; ++u; }
    pair_su[t] = (uint16_t)((t - tu) | (u << 8));
  }
  __syncthreads();                                   // (before any wave leaves)
  const NmsFrame fr = *frame;
  // the cells the frame kernel laid out are [0, ggx * ggy), ggx = 2^(15 - gshx): the launch is sized for the largest
  // grid, the chunks for the one that exists (the headline: 128 x 128 of at most 512 x 512 cells); the groups of a
  // grouped call are spread over the whole table
  const uint32_t used = (uint32_t)fr.ggx * (uint32_t)fr.ggy;
  const uint32_t cells = (GROUPED || used >= ncells) ? ncells : used;
  const uint32_t mask = ncells - 1u;
  const uint32_t cl = grid_chunk_log2(cells);
  const uint32_t chunks = (cells + (1u << cl) - 1u) >> cl;
  const uint32_t nover_all = ctr->grid_over;                      // (written by the insert launch)
  const uint32_t nover = nover_all < over_cap ? nover_all : over_cap;
  const uint32_t nbig_all = ctr->grid_big;
  const uint32_t nbig = nbig_all < big_cap ? nbig_all : big_cap;
  // (the insert launch's verdict, requested WITH the words above: an agent-scope load behind them was a second round trip at
  //  the head of every wave; what THIS launch adds to the word nobody here needs to see -- the host reads it in the record)
  const uint32_t gave_up = ctr->grid_fail;
  if (gw >= chunks && gw >= nover && nbig == 0u) return;
  if (gave_up != 0u) return;                                      // the call is repeated on the sorted route
  GP_STAMP(1);
  uint2 *q = queue[wave];
  uint4 *ew = ent[wave];
  int qn = 0;
  const int shard = (int)(gw & (kShards - 1));
  DirectSink sink;
  sink.pr.thr = thr; sink.pr.cpx = 0.0f; sink.pr.cpy = 0.0f;
  sink.per_box_thr = per_box_thr != 0;
  sink.flags = flags;
  sink.iv = nullptr; sink.ar = nullptr; sink.rec = rec; sink.klo = nullptr;
  sink.eseg = edges + (size_t)shard * segcap;
  sink.segcap = segcap;
  sink.ccnt = ccnts + (size_t)shard * kCandStride;
  sink.ecnt = ccnts + (size_t)(kShards + shard) * kCandStride;
  sink.cnt = cnt;
  Prune tp;                                       // bounds applied to the table entries (off for per-box thresholds: the records hold those)
  tp.thr = per_box_thr ? 0.0f : thr; tp.cpx = fr.cpx; tp.cpy = fr.cpy;
  if (thr_table) {
    // thresholds from a per-class table (NmsSource mode 2): every pair's threshold is one of the table's or the call's
    // (src_thr) -- the SMALLEST of them bounds the entries (the records decide with the pair's own)
    float m = thr;
    bool ok = thr > 0.0f;
    for (int k = lane; k < thr_count; k += 64) { const float v = thr_table[k]; ok = ok && v > 0.0f; m = v < m ? v : m; }
    for (int o = 32; o > 0; o >>= 1) { const float v = __shfl_xor(m, o); m = v < m ? v : m; }
    tp.thr = __ballot(!ok) == 0ull ? m : 0.0f;     // (a threshold <= 0 or NaN: no bound)
  }
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const int s32 = lane & 31, hi = lane >> 5;
  const int xbits = 15 - fr.gshx;
  // a candidate of THIS cell: one tag, the intervals meet and the cell holds the lower corner of their intersection (the
  // one cell of all that both boxes touch where the pair is reported); with ONE threshold, the two bounds on the entries
  auto pair_here = [&](const uint4 &ea, const uint4 &eb, uint32_t cell) {
    const uint32_t axl = ea.x & 0x7FFFu, ayl = (ea.x >> 16) & 0x7FFFu;
    const uint32_t bxl = eb.x & 0x7FFFu, byl = (eb.x >> 16) & 0x7FFFu;
    const uint32_t ox = (axl > bxl ? axl : bxl) >> fr.gshx, oy = (ayl > byl ? ayl : byl) >> fr.gshy;
    uint32_t owner = (oy << xbits) | ox;
    bool cnd = qdisjoint(ea.x, ea.y, eb.x, eb.y) == 0u;
    if (GROUPED) {
      owner = (owner + grid_tag_offset(ea.w & 0xFFFFu, mask)) & mask;
      cnd = cnd && ((ea.w ^ eb.w) & 0xFFFFu) == 0u;
    }
    cnd = cnd && owner == cell;
    if (tp.thr > 0.0f) {
      // the entries hold lower bounds of the areas (the upper 16 bits of the float): the smaller area is at most
      // 1.0079 x its bound, the sum at least the sum of the bounds -- both bounds stay on the safe side
      const float aa = __uint_as_float(ea.w & 0xFFFF0000u), ab = __uint_as_float(eb.w & 0xFFFF0000u);
      // (an interval that touches the first or last cell of the quantisation may have been CLAMPED into the frame: its
      // cells then say nothing about its extent -- as in the sorted route's scan)
      const bool edge = qedge2(ea.x, ea.y, eb.x, eb.y);
      cnd = cnd && !(min_sel(aa, ab) * 1.0079f < (tp.thr * max_sel(aa, ab)) * 0.999999f) &&           // (prune_area on the bounds)
            (edge || prune_aabb(tp, ea.x, ea.y, aa, eb.x, eb.y, ab));
    }
    return cnd;
  };
#ifdef OBB_GPROBE_STAMPS
  unsigned long long acc_flush = 0ull, acc_over = 0ull, n_flush = 0ull;      // (clocks inside the in-loop flushes / the overflow pass)
#define GP_ACC_BEGIN() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define GP_ACC_END(acc_) (acc_) += __builtin_amdgcn_s_memtime() - t0_
#else
#define GP_ACC_BEGIN() do { } while (0)
#define GP_ACC_END(acc_) do { } while (0)
#endif
  auto push = [&](bool cnd, uint32_t ia, uint32_t ib) {
    const uint64_t mcnd = __ballot(cnd);
    if (mcnd) {
      if (cnd) q[qn + __popcll(mcnd & lt_mask)] = make_uint2(ia, ib);
      qn += __popcll(mcnd);
      if (qn >= 64) {
        qn -= 64;
        __builtin_amdgcn_wave_barrier();
        GP_ACC_BEGIN();
        sink.flush(q + qn, 64, lane);
        GP_ACC_END(acc_flush);
#ifdef OBB_GPROBE_STAMPS
        ++n_flush;
#endif
      }
    }
  };
  for (uint32_t chunk = gw; chunk < chunks; chunk += nw) {
    const uint32_t cell0 = chunk << cl;
    uint32_t mc = 0;
    if (lane < (1 << cl) && cell0 + (uint32_t)lane < cells) {
      const uint32_t c = gcnt[cell0 + lane];
      mc = c < (uint32_t)kGridCap ? c : (uint32_t)kGridCap;
    }
    int m[kGridCpwMax], O[kGridCpwMax + 1];
    O[0] = 0;
#pragma unroll
    for (int k = 0; k < kGridCpwMax; ++k) {
      m[k] = __builtin_amdgcn_readlane((int)mc, k);
      O[k + 1] = O[k] + ((m[k] * (m[k] - 1)) >> 1);
    }
    const int total = O[kGridCpwMax];
    if (total == 0) continue;                      // (wave-uniform) no pair in the chunk
    if (lane < kGridCpwMax) {
      int of = 0;
#pragma unroll
      for (int k = 1; k < kGridCpwMax; ++k) of = lane >= k ? O[k] : of;
      pair_first[wave][lane] = of;
    }
    uint4 e[kGridCpwMax / 2];
#pragma unroll
    for (int r = 0; r < kGridCpwMax / 2; ++r) {
      const int mm = m[2 * r] + hi * (m[2 * r + 1] - m[2 * r]);   // (a select of two array elements becomes an indexed read of a scratch copy)
      e[r] = make_uint4(0u, 0u, 0u, 0u);
      if (s32 < mm) e[r] = gtab[(size_t)(cell0 + 2 * r + hi) * kGridCap + s32];
    }
#pragma unroll
    for (int r = 0; r < kGridCpwMax / 2; ++r) ew[r * 64 + lane] = e[r];
    __builtin_amdgcn_wave_barrier();
    GP_STAMP(2);
    for (int p0 = 0; p0 < total; p0 += 64) {
      const int p = p0 + lane;
      const bool valid = p < total;
      int c = 0;
#pragma unroll
      for (int k = 1; k < kGridCpwMax; ++k) c += (p >= O[k]) ? 1 : 0;
      const int tf = p - pair_first[wave][c];         // (c <= 7 whatever p: read by every lane, no branch)
      const int t = valid ? tf : 0;
      const int su = (int)pair_su[t];
      const int sidx = su & 255, u = su >> 8;
      const uint4 ea = ew[c * kGridCap + sidx], eb = ew[c * kGridCap + u];
      const bool cnd = valid && pair_here(ea, eb, cell0 + (uint32_t)c);
      push(cnd, ea.z, eb.z);
    }
    __builtin_amdgcn_wave_barrier();               // (the next chunk's entries overwrite `ew`)
  }
  // ---- the overflow list: a wave per entry that found its cell full -- against the 32 entries the cell holds, and
  // against the later entries of the list that belong to the same cell (every pair of a crowded cell once)
  if (nover > 0u) {
    GP_ACC_BEGIN();
    const uint32_t *__restrict__ gover_cell = reinterpret_cast<const uint32_t *>(gover + kGridOverMax);
    for (uint32_t o = gw; o < nover; o += nw) {
      const uint4 eo = gover[o];
      const uint32_t cell = gover_cell[o];
      {
        uint4 et = make_uint4(kSentLo, kSentHi, 0u, 0u);
        if (lane < kGridCap) et = gtab[(size_t)cell * kGridCap + lane];
        push(lane < kGridCap && pair_here(eo, et, cell), eo.z, et.z);
      }
      for (uint32_t j0 = o + 1u; j0 < nover; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        const bool same = j < nover && gover_cell[j] == cell;
        if (__ballot(same) == 0ull) continue;
        uint4 ej = make_uint4(kSentLo, kSentHi, 0u, 0u);
        if (same) ej = gover[j];
        push(same && pair_here(eo, ej, cell), eo.z, ej.z);
      }
    }
    GP_ACC_END(acc_over);
  }
  // ---- the big list: every box that touches more cells than an entry list may name, against EVERY box of the call, 64
  // per step -- their packed intervals are rebuilt from the records (grid_interval: the insert kernel's expressions), so
  // the common call stores nothing for this pass; two big boxes meet once, in the step of the one with the lower number
  if (nbig > 0u) {
    const uint32_t nblk = (n + 63u) >> 6;
    const unsigned long long items = (unsigned long long)nbig * nblk;
    auto packed = [&](uint32_t x, uint2 &o, bool &is_big) {
      const float4 *r4 = reinterpret_cast<const float4 *>(rec + x);
      const float4 f = r4[0];                     // ox, oy, ex, ey
      bool wellformed = true;
      if (flags & kFlagHbb) { const float4 c = r4[1]; wellformed = ((c.z - c.x) + 1.0f > 0.0f) && ((c.w - c.y) + 1.0f > 0.0f); }   // (hbb_record)
      uint32_t cx0, cx1, cy0, cy1;
      const bool in = grid_interval(f.x, f.z, f.y, f.w, wellformed, fr, o, cx0, cx1, cy0, cy1);
      is_big = in && (cx1 - cx0 + 1u) * (cy1 - cy0 + 1u) > (uint32_t)kGridMaxCells;
    };
    for (unsigned long long it = gw; it < items; it += nw) {
      const uint32_t bi = (uint32_t)(it / nblk), blk = (uint32_t)(it - (unsigned long long)bi * nblk);
      const uint32_t B = gbig[bi];
      uint2 qB;
      bool bigB;
      packed(B, qB, bigB);
      const float aB = ar[B];
      const uint32_t x = blk * 64u + (uint32_t)lane;
      bool cnd = false;
      if (x < n && x != B) {
        uint2 qx;
        bool bigx;
        packed(x, qx, bigx);
        cnd = qdisjoint(qB.x, qB.y, qx.x, qx.y) == 0u && (!bigx || B < x);
        if (cnd && tp.thr > 0.0f) {
          const float ax = ar[x];
          cnd = !(min_sel(aB, ax) < (tp.thr * max_sel(aB, ax)) * 0.999999f) &&
                (qedge2(qB.x, qB.y, qx.x, qx.y) || prune_aabb(tp, qB.x, qB.y, aB, qx.x, qx.y, ax));
        }
      }
      push(cnd, B, x);
    }
  }
  GP_STAMP(3);
  if (qn > 0) {
    __builtin_amdgcn_wave_barrier();
    sink.flush(q, qn, lane);
  }
  GP_STAMP(7);
  GP_STAMP(9);
#ifdef OBB_GPROBE_STAMPS
  if (lane == 0 && gw < 32768u && chunks > nw) {      // (slots 4 .. 6 of a multi-chunk wave: totals instead of the last flush's stamps)
    g_gprobe_stamps[gw * 10 + 4] = acc_flush; g_gprobe_stamps[gw * 10 + 5] = acc_over; g_gprobe_stamps[gw * 10 + 6] = n_flush;
  }
#endif
}

#ifdef OBB_GPROBE_STAMPS
}  // namespace obb
extern "C" __attribute__((visibility("default"))) int obb_debug_gprobe_stamps(unsigned long long *host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_gprobe_stamps), sizeof(unsigned long long) * (size_t)words);
}
namespace obb {
#endif

// ---- 7: clip kernel -----------------------------------------------------------------------
// One candidate pair per lane.  The edge points from the box that PRECEDES in the
// greedy order (smaller score key, then smaller index) to the other one; the clip
// always takes the preceding box as the subject A, exactly like the i-before-j
// loop of nms_cpu.cpp:35-58.  grid = (blocks per shard, shards).
__global__ __launch_bounds__(256) void nms_clip_kernel(const uint2 *__restrict__ cands,
                                                       unsigned long long segcap,
                                                       const QuadRec *__restrict__ rec, int flags,
                                                       uint2 *__restrict__ edges,
                                                       NmsCounters *__restrict__ ctr,
                                                       unsigned long long *__restrict__ ccnts,
                                                       int *__restrict__ cnt, const uint32_t *__restrict__ klo) {
  const int lane = threadIdx.x & 63;
  const int shard = blockIdx.y;
  const uint2 *cand = cands + (size_t)shard * segcap;
  uint2 *eseg = edges + (size_t)shard * segcap;                       // this shard's edge segment
  unsigned long long *ecnt = ccnts + (size_t)(kShards + shard) * kCandStride;
  // the first candidate of this lane is loaded together with the shard's count (one memory round
  // trip instead of two on a latency-bound kernel; the slot exists whatever the count says)
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
  uint2 pr = make_uint2(0u, 0u);
  if (first + lane < segcap) pr = cand[first + lane];
  unsigned long long total = ccnts[(size_t)shard * kCandStride];
  if (total > segcap) total = segcap;  // overflow is reported by the host
  // whole waves iterate together so that the ballot below sees all 64 lanes
  for (unsigned long long base = first; base < total; base += stride) {
    const unsigned long long e = base + lane;
    bool hit = false;
    uint32_t src = 0, dst = 0;
    if (e < total) {
      if (base != first) pr = cand[e];
      hit = clip_pair(rec, pr, flags, src, dst, klo);
    }
    push_edges(hit, src, dst, eseg, segcap, ecnt, cnt, lane);
  }
}

// ---- 8: resolve ---------------------------------------------------------------
// One word per box: the number of LIVE incoming edges (bits 0..29; set by the clip's
// atomicAdd) and a SUPPRESSED flag (bit 30).  A box is KEPT iff its word is zero: no kept
// predecessor can reach it any more.  Both transitions are fire-and-forget:
//   source kept       ->  word[target] = SUPPRESSED | all count bits   (an atomic store, mark_suppressed)
//   source suppressed ->  atomicSub(word[target], 1)
// and each word only ever moves towards its final value, so the iteration may run
// ASYNCHRONOUSLY: 64 x P workgroups, P per edge shard, each walking its own part of the segment
// (compacted in place) until it is empty, reading the others' progress through agent-scope loads.
// A stale read only postpones a decision to the next sweep.  No grid barrier: a workgroup
// whose edges wait for other shards just sweeps again (every sweep retires at least the
// edge with the globally earliest undecided source once that shard has run) -- which is why ALL the
// workgroups must be resident together (resolve_parts() below bounds the grid by the device's occupancy;
// a workgroup that reaches its bound of sweeps publishes kStalled and the host returns OBB_ESTALLED).
// One workgroup = one CU sustains about a third of a scattered memory request per clock -- the
// single-workgroup version of this kernel needed 170 us for 3*10^4 edges.
constexpr int kSuppressedBit = 1 << 30;
// Instrumented build only (-DOBB_RESOLVE_STAMPS -> libobbops_rsst.so, tools/exp_resolve_stamps.py): per shard and sweep the
// live edges at its start and the 100 MHz wall clock; row 0 of a shard: entry clock, exit clock, sweeps
#ifdef OBB_RESOLVE_STAMPS
__device__ unsigned long long g_resolve_stamps[64 * 64 * 2];
#define RS_NOTE(sweep, a, b) do { if (threadIdx.x == 0 && (sweep) < 64 && part == 0) { g_resolve_stamps[(shard * 64 + (sweep)) * 2] = (unsigned long long)(a); g_resolve_stamps[(shard * 64 + (sweep)) * 2 + 1] = (unsigned long long)(b); } } while (0)   /* (part 0 of every shard) */
#else
#define RS_NOTE(sweep, a, b) do { } while (0)
#endif
constexpr unsigned int kCountValid = 0x80000000u;   // flag of a handed-on count (nms_compact_kernel, the fused compaction)
constexpr int kResolveThreads = 512;          // sorted / grid route: 64 (x parts) workgroups share the edges; three of them fit a CU
constexpr int kResolveThreadsCompact = 1024;  // direct route: usually ONE workgroup does it all
constexpr int kResolveParts = 8;              // workgroups per shard in calls without the fused compaction (n >= 2^18)
constexpr long long kSoloEdges = 2048;      // up to here one workgroup resolves alone (direct route)
constexpr long long kStalled = -1;           // `rounds` of a call whose resolve stopped at its bound (nms_verdict: OBB_ESTALLED)
static int resolve_parts();                   // (OBB_TUNE_NMS_RESOLVE_PARTS, bounded by what the device keeps resident: below the kernel)

__device__ __forceinline__ int ld_word(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool word_kept(int w) { return w == 0; }
// word of box i (i < 2^24: a 32-bit byte offset beside the scalar base -- no 64-bit address registers per request in flight)
__device__ __forceinline__ int *word_at(int *cnt, uint32_t i) {
  return reinterpret_cast<int *>(reinterpret_cast<char *>(cnt) + (size_t)(uint32_t)(i << 2));
}
// "source kept -> target suppressed" as an atomic STORE, not a read-modify-write: the flag with all the count bits below
// it set.  The count of a suppressed word is never read again, a decrement that races with the store (at most one per
// incoming edge, < 2^24) cannot borrow its way up to bit 30, and a store is a cheaper fabric request than a memory-side
// read-modify-write (config-5 merge, 480 000 suppressions: resolve 39 -> 28 us).
__device__ __forceinline__ void mark_suppressed(int *p) {
  __hip_atomic_store(p, 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// COMPACT: the workgroup that finishes LAST also writes the ascending keep indices and the
// host info (medium n: two more launches would cost more than the serial compaction).
template <bool COMPACT>
__global__ __launch_bounds__(COMPACT ? kResolveThreadsCompact : kResolveThreads) void nms_resolve_kernel(uint2 *__restrict__ edges,
                                                           unsigned long long segcap,
                                                           int *__restrict__ cnt,
                                                           NmsCounters *__restrict__ ctr,
                                                           const unsigned long long *__restrict__ ccnts,
                                                           int64_t n, int64_t *__restrict__ keep_out,
                                                           NmsHostInfo *__restrict__ hinfo, int coop,
                                                           unsigned int *__restrict__ blockcnt,
                                                           const unsigned int *__restrict__ ucnts, int tuning_tail,
                                                           NmsGridCtr *__restrict__ gctr, const RetainTail rt) {
  constexpr int kThreads = COMPACT ? kResolveThreadsCompact : kResolveThreads;
  constexpr int kW = kThreads / 64;
  __shared__ int overflow;
  __shared__ long long pre[kShards + 1];   // virtual start of every edge segment
  __shared__ int is_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // a retained call whose workspace somebody else wrote since its predecessor (NmsRetain): the insert and probe launches have
  // left without touching anything, this one only says so -- the host seeds the workspace again
  if constexpr (!COMPACT) if (rt.hdr_magic && rt.expect && *rt.hdr_magic != rt.expect) {
    if (blockIdx.x == 0 && tid == 0) {
      hinfo->kept = 0; hinfo->edges = 0ull; hinfo->rounds = 0; hinfo->cand_total = 0ull; hinfo->cand_worst = 0ull;
      hinfo->heavy_units = 0u; hinfo->light_units = 0u; hinfo->route = kRouteUnseeded; hinfo->unit_worst = 0u;
    }
    return;
  }
  // tail of a call on a retained workspace: this workgroup's share of the cell counters is cleared as the LAST thing it does
  // (the probe launch has read them; behind the keep list and the record, so that nothing of the call waits for these stores)
  auto retain_clear = [&]() __attribute__((always_inline)) {
    if constexpr (!COMPACT) if (rt.gcnt)
      for (size_t z = (size_t)blockIdx.x * kThreads + tid; z < rt.gcnt_vecs; z += (size_t)gridDim.x * kThreads) rt.gcnt[z] = make_uint4(0u, 0u, 0u, 0u);
  };
  // large calls (no fused compaction): gridDim.x = kShards * P workgroups, P of them share a shard's segment -- one CU
  // issues about a third of a scattered request per clock, and a segment of 13 000 edges was 58 us of first sweep
  const int parts = (int)gridDim.x / kShards, shard = (int)blockIdx.x / parts, part = (int)blockIdx.x - shard * parts;
  // fused compaction: the sums of the call's result record are final when this launch starts (the earlier launches wrote
  // them) -- wave 1 gathers them HERE, beside wave 0's prefix, instead of five loads and thirty shuffles behind the last
  // slice's writes at the very end of the call (~1 us of every call's tail)
  __shared__ unsigned long long s_info[2];
  __shared__ unsigned int s_units[4];
  if constexpr (!COMPACT) if (coop && wave == 1) {
    unsigned long long v = ccnts[(size_t)lane * kCandStride], mx = v;
    const unsigned long long ev = ccnts[(size_t)(kShards + lane) * kCandStride];
    mx = ev > mx ? ev : mx;
    v += ccnts[(size_t)(2 * kShards + lane) * kCandStride];
    unsigned int uh = ucnts[lane * kUnitStride], ul = ucnts[(kShards + lane) * kUnitStride], uw = uh + ul;
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_xor(v, o);
      const unsigned long long tt = __shfl_xor(mx, o);
      mx = tt > mx ? tt : mx;
      uh += __shfl_xor(uh, o); ul += __shfl_xor(ul, o);
      const unsigned int tw = __shfl_xor(uw, o);
      uw = tw > uw ? tw : uw;
    }
    if (lane == 0) {
      s_info[0] = v; s_info[1] = mx; s_units[0] = uh; s_units[1] = ul; s_units[2] = uw;
      s_units[3] = gctr ? __hip_atomic_load(&gctr->grid_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // (set by the insert launch only; null: not the grid route)
    }
  }
  if (wave == 0) {  // 64 lanes = 64 shards
    const unsigned long long c = ccnts[(size_t)(kShards + lane) * kCandStride];
    const unsigned long long cc = ccnts[(size_t)lane * kCandStride];
    long long inc = (long long)c;
    for (int o = 1; o < 64; o <<= 1) {
      const long long t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    pre[lane] = inc - (long long)c;
    const uint64_t ovm = __ballot((c > segcap) || (cc > segcap));
    if (lane == 63) {
      pre[kShards] = inc;
      overflow = ovm != 0;
      if (blockIdx.x == 0) ctr->edges = (unsigned long long)inc;
    }
  }
  __syncthreads();
  const long long total_edges = pre[kShards];
  // SOLO: with few edges (the usual case of the direct route) workgroup 0 walks all the
  // segments itself and the other 63 leave at once -- no hand-over between workgroups
  const bool solo = COMPACT && total_edges <= kSoloEdges;
  if (solo && shard != 0) return;
  // an overflowed candidate / edge list is incomplete (the words count edges that were
  // not stored): nothing to resolve, the host reports OBB_EWORKSPACE
  uint2 *seg = solo ? edges : edges + (size_t)shard * segcap;
  long long live = overflow ? 0 : (solo ? total_edges : pre[shard + 1] - pre[shard]);
  if (parts > 1) {                           // this workgroup's part of the segment (compacted inside its own range)
    const long long per = (live + parts - 1) / parts, lo = per * part < live ? per * part : live;
    seg += lo;
    live = live - lo < per ? live - lo : per;
  }
  bool segmented = solo;   // sweep 1 of SOLO reads the 64 segments, later ones the compacted front
  long long rounds = 0;
  bool stalled = false;
  constexpr int kPer = 4;  // edges per thread and chunk: their dependent loads overlap
  // (a segment goes to the register tail when it fits it AFTER a sweep, or at once when it is tiny: the first sweep's
  //  cached loads are cheaper than the tail's polling, which every polling workgroup pays for at the fabric)
  constexpr int kTailFirst = 512;
  const bool kTail = tuning_tail & 1, first_mode = !(tuning_tail & 2);
  if (tuning_tail & 4) stalled = true;      // (test hook: the host's OBB_ESTALLED path without a starved device)
  __shared__ int wsum4[kPer][kW];
  __shared__ int s_tail_rounds;
  RS_NOTE(0, wall_clock64(), live);
  while (live > 0) {
    RS_NOTE(1 + (int)rounds, live, wall_clock64());
    if (kTail && live <= (rounds == 0 ? (long long)kTailFirst : (long long)kThreads * kPer)) {
      // ---- the TAIL: what is left fits the threads' registers (<= kPer edges each).  No write-back of the survivors: a
      // thread re-reads the two words of ITS undecided edges until the workgroup has retired them all -- a sweep is one
      // load round trip and a barrier instead of edge reload + words + two barriers + compaction (the last sweeps of a call
      // walk suppression chains a few edges at a time).  Progress is the other threads' and shards' atomics, as before; a
      // stale read still only postpones the decision.  (The barrier per sweep is not optional: with every wave polling on
      // its own, a 2000-box chain in ONE workgroup stopped advancing after its first two waves -- 4 M polls, wrong keeps.)
      uint2 ed[kPer];
      bool pend[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const long long e = (long long)k * kThreads + tid;
        pend[k] = e < live;
        ed[k] = make_uint2(0u, 0u);
        if (pend[k]) {
          long long phys = e;
          if (segmented) {
            int lo = 0, hi = kShards - 1;
            while (lo < hi) {
              const int mid = (lo + hi + 1) >> 1;
              if (pre[mid] <= e) lo = mid; else hi = mid - 1;
            }
            phys = (long long)lo * (long long)segcap + (e - pre[lo]);
          }
          ed[k] = seg[phys];
        }
      }
      if (tid == 0) s_tail_rounds = 0;
      __syncthreads();
      int it = 0;
      bool any = pend[0] | pend[1] | pend[2] | pend[3];
      static_assert(kPer == 4, "the tail's pending flags");
      while (__syncthreads_or(any ? 1 : 0) && it < (1 << 22)) {
        int ws[kPer], wd[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {       // (unconditional, clamped: the loads of a sweep travel together)
          wd[k] = ld_word(word_at(cnt, ed[k].y));
          ws[k] = ld_word(word_at(cnt, ed[k].x));
        }
        any = false;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
          if (pend[k]) {
            if (wd[k] & kSuppressedBit) pend[k] = false;
            else if (ws[k] & kSuppressedBit) { atomicSub(word_at(cnt, ed[k].y), 1); pend[k] = false; }
            else if (word_kept(ws[k])) { mark_suppressed(word_at(cnt, ed[k].y)); pend[k] = false; }
            else any = true;
          }
        }
        ++it;
      }
      if (lane == 0 || it > 0) atomicMax(&s_tail_rounds, it);
      __syncthreads();
      rounds += s_tail_rounds;
      if (s_tail_rounds >= (1 << 22)) stalled = true;
      live = 0;
      break;
    }
    long long w = 0;
    for (long long base = 0; base < live; base += kThreads * kPer) {
      uint2 ed[kPer];
      bool have[kPer], keep[kPer];
      int ws[kPer], wd[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const long long e = base + (long long)k * kThreads + tid;
        have[k] = e < live;
        keep[k] = false;
        ed[k] = make_uint2(0u, 0u);
        if (have[k]) {
          long long phys = e;
          if (segmented) {  // virtual index -> (segment, offset); physical >= virtual, so the
                            // in-place compaction below never overwrites an unread edge
            int lo = 0, hi = kShards - 1;
            while (lo < hi) {
              const int mid = (lo + hi + 1) >> 1;
              if (pre[mid] <= e) lo = mid; else hi = mid - 1;
            }
            phys = (long long)lo * (long long)segcap + (e - pre[lo]);
          }
          ed[k] = seg[phys];
        }
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        ws[k] = 0; wd[k] = kSuppressedBit;
        if (have[k]) {
          if (first_mode && rounds == 0) {
            // the FIRST sweep reads the words as the launch found them: a cached load of the source (a zero is final:
            // counts only fall and a zero word has no edge left to suppress it; anything else waits for sweep 2) and no
            // look at the target (nothing is suppressed yet; a store to a target another shard got first is repeated).
            // Agent-scope loads go past the L2 to the fabric, and the fabric -- not the 64 CUs -- bounds a large call.
            wd[k] = 0; ws[k] = *word_at(cnt, ed[k].x);
          } else { wd[k] = ld_word(word_at(cnt, ed[k].y)); ws[k] = ld_word(word_at(cnt, ed[k].x)); }
        }
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        if (have[k] && !(wd[k] & kSuppressedBit)) {   // (a suppressed target: the edge is moot)
          if (ws[k] & kSuppressedBit) atomicSub(word_at(cnt, ed[k].y), 1);
          else if (word_kept(ws[k])) mark_suppressed(word_at(cnt, ed[k].y));
          else keep[k] = true;
        }
      }
      // compact the survivors to the front of the segment (any order is fine; reads of a
      // chunk complete before its writes and the write cursor never passes the chunk)
      uint64_t mk[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        mk[k] = __ballot(keep[k]);
        if (lane == 0) wsum4[k][wave] = __popcll(mk[k]);
      }
      __syncthreads();
      int run = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        int off = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < kW; ++j) {
          const int v = wsum4[k][j];
          off += (j < wave) ? v : 0;
          tot += v;
        }
        if (keep[k]) seg[w + run + off + __popcll(mk[k] & ((1ull << lane) - 1ull))] = ed[k];
        run += tot;
      }
      w += run;
      __syncthreads();
    }
    live = w;
    segmented = false;
    ++rounds;
    if (rounds > (1ll << 22)) { stalled = true; break; }   // corrupted state, or workgroups this one waits for never ran: stop instead of spinning forever
  }
  RS_NOTE(63, wall_clock64(), rounds);
  // (a stalled resolve publishes all ones = kStalled: the maximum of every shard's word, so it survives)
  if (tid == 0) atomicMax(reinterpret_cast<unsigned long long *>(&ctr->rounds), stalled ? ~0ull : (unsigned long long)rounds);
  if constexpr (!COMPACT) if (coop) {
    // ---- fused compaction (round 4: the sorted route's nms_compact_kernel launch and its boundary, n < 2^18).  The 64
    // workgroups of this launch are co-resident by size (64 x 512 threads), so the shards that are done WAIT for the last
    // one -- on one word, with a bound -- and then compact n / 64 words each (counts handed on through flagged words, as in
    // nms_compact_kernel).  No deadlock whatever the scheduler does: a waiter whose bound expires raises a flag in the
    // arrival word -- possible only while arrivals are missing -- and from then on every shard leaves after its resolve
    // except the last one to do so, which by then sees every final word and writes the whole keep list alone (below).
    __shared__ int s_go;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's atomics are acknowledged before the ticket is taken)
    __syncthreads();
    if (tid == 0) {
      // `done`: arrivals in the low bits, kGave = a waiter's bound expired.  An arrival is a fire-and-forget add and every
      // shard -- the last one too -- then READS the word until it shows all arrivals (go) or the flag: one load round trip
      // behind the last add where a returning ticket and a compare-and-swap of a second word (two returning atomics in a
      // row on the last shard's path, ~2 us each) used to be.  The flag can only be set while arrivals are missing (a
      // compare-and-swap against the word just read), so "all arrived, no flag" is final for everybody who sees it; with
      // the flag up every shard registers in `state` (a returning add: the rare path) and the LAST to register -- all the
      // others have resolved and left by then -- writes the whole keep list alone.
      constexpr unsigned int kGave = 0x80000000u;
      unsigned int st = 0u;
      (void)__hip_atomic_fetch_add(&ctr->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint64_t t0 = wall_clock64();
      for (;;) {
        const unsigned int w = __hip_atomic_load(&ctr->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w & kGave) { st = 2u; break; }
        if (w == gridDim.x) { st = 1u; break; }
        if (wall_clock64() - t0 >= (coop == 2 ? 0ull : 25000000ull)) {     // 0.25 s of the 100 MHz wall clock (coop 2: a test hook, at once)
          unsigned int expect = w;
          if (__hip_atomic_compare_exchange_strong(&ctr->done, &expect, w | kGave, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { st = 2u; break; }
          continue;                                                          // (the word moved: look again)
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (st == 2u) {
        const unsigned int reg = __hip_atomic_fetch_add(&ctr->state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (reg == gridDim.x - 1) st = 3u;         // everybody else has resolved, registered and left: this workgroup compacts everything
      }
      s_go = (int)st;
    }
    __syncthreads();
    const int go = s_go;
    if (go == 2) { retain_clear(); return; }
    // one pass: a thread requests its (up to) 16 words of the slice together -- they are agent-scope loads, a fabric
    // round trip each -- keeps the verdicts in registers, the 16 x 4 (row, wave) counts are scanned by one wave
    constexpr int kR = 64 / kW, kSlice = kR * kThreads;     // 4096 words: a slice of n < 2^18 boxes over 64 shards
    static_assert(kR * kW == 64, "one wave scans the (row, wave) counts");
    __shared__ int csum[kR][kW];
    __shared__ long long s_base;
    const int64_t per = go == 1 ? (n + kShards - 1) / kShards : n;
    const int64_t lo = go == 1 ? (int64_t)shard * per : 0, hi = (lo + per < n) ? lo + per : n;
    __shared__ int s_tot;
    long long kbase = 0;
    bool first_chunk = true;
    for (int64_t c0 = lo; first_chunk || c0 < hi; c0 += kSlice) {
      bool f[kR];
      uint64_t mk[kR];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int64_t i = c0 + (int64_t)r * kThreads + tid;
        const int wv = ld_word(cnt + (i < hi ? i : (n > 0 ? n - 1 : 0)));    // (unconditional: all rows in flight together)
        f[r] = (i < hi) && word_kept(wv);
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        mk[r] = __ballot(f[r]);
        if (lane == 0) csum[r][wave] = __popcll(mk[r]);
      }
      __syncthreads();
      if (wave == 0) {                      // exclusive scan of the 64 (row, wave) counts, row-major
        const int v = (&csum[0][0])[lane];
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(inc, o); if (lane >= o) inc += t2; }
        (&csum[0][0])[lane] = inc - v;
        const int tot = __shfl(inc, 63);
        long long base = kbase;
        if (first_chunk && go == 1) {       // hand the slice's count on, take the predecessors' (at most 63)
          if (lane == 0) __hip_atomic_store(&blockcnt[1 + shard], (unsigned int)tot | kCountValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          long long part = 0;
          if (lane < shard) {
            unsigned int v2;
            do { v2 = __hip_atomic_load(&blockcnt[1 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v2 & kCountValid));
            part = (long long)(v2 & ~kCountValid);
          }
          for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
          base = part;
        }
        if (lane == 0) { s_base = base; s_tot = tot; }
      }
      __syncthreads();
      const long long base = s_base;
#pragma unroll
      for (int r = 0; r < kR; ++r)
        if (f[r]) keep_out[base + csum[r][wave] + __popcll(mk[r] & ((1ull << lane) - 1ull))] = c0 + (int64_t)r * kThreads + tid;
      kbase = base + s_tot;
      first_chunk = false;
      __syncthreads();                      // (csum / s_base are reused by the next chunk)
    }
    if ((go == 3 || shard == kShards - 1) && tid == 0) {   // the last slice knows K: the call's result record
      hinfo->kept = kbase; hinfo->edges = (unsigned long long)total_edges;
      hinfo->rounds = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&ctr->rounds), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hinfo->cand_total = s_info[0]; hinfo->cand_worst = s_info[1];
      hinfo->heavy_units = s_units[0]; hinfo->light_units = s_units[1];
      const unsigned int gf = s_units[3];
      hinfo->route = gf ? (3u | (gf << 8)) : 2u;   // 3: repeat without the grid (bits 8..: which skip hint, nms_ginsert_kernel)
      hinfo->unit_worst = s_units[2];
      if (rt.hdr_magic) {
        // retained workspace: the three words the NEXT call's insert launch counts in (every workgroup of this launch has
        // read grid_fail: they have all arrived, or left), and the header now names that call
        gctr->grid_over = 0u; gctr->grid_big = 0u; gctr->grid_fail = 0u;
        *rt.hdr_magic = rt.next;
      }
    }
    retain_clear();
    return;
  }
  if (COMPACT) {
    if (!solo) {
      // the last workgroup to get here sees every final word: the payload is atomics only
      // (performed at the coherence point, nothing dirty in a cache), so every wave waits
      // for its own to be acknowledged -- __syncthreads alone only waits for LDS -- and
      // then ONE thread takes the ticket.  (An agent-scope release fence in every wave
      // costs a cache write-back each: tens of microseconds.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        is_last = __hip_atomic_fetch_add(&ctr->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      __syncthreads();
      if (!is_last) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    constexpr int kC = 4;   // boxes per thread and sweep (independent loads in flight)
    __shared__ int csum[kC][kW];
    long long kbase = 0;
    for (int64_t c0 = 0; c0 < n; c0 += kThreads * kC) {
      bool f[kC];
      uint64_t mk[kC];
#pragma unroll
      for (int k = 0; k < kC; ++k) {
        const int64_t i = c0 + (int64_t)k * kThreads + tid;
        f[k] = (i < n) && word_kept(ld_word(cnt + i));
      }
#pragma unroll
      for (int k = 0; k < kC; ++k) {
        mk[k] = __ballot(f[k]);
        if (lane == 0) csum[k][wave] = __popcll(mk[k]);
      }
      __syncthreads();
      int run = 0;
#pragma unroll
      for (int k = 0; k < kC; ++k) {
        int off = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < kW; ++j) {
          const int v = csum[k][j];
          off += (j < wave) ? v : 0;
          tot += v;
        }
        if (f[k]) keep_out[kbase + run + off + __popcll(mk[k] & ((1ull << lane) - 1ull))] = c0 + (int64_t)k * kThreads + tid;
        run += tot;
      }
      kbase += run;
      __syncthreads();
    }
    if (wave == 0) {
      unsigned long long v = ccnts[(size_t)lane * kCandStride], mx = v;
      const unsigned long long ev = ccnts[(size_t)(kShards + lane) * kCandStride];
      mx = ev > mx ? ev : mx;   // fullest segment of either kind
      for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o);
        const unsigned long long t = __shfl_xor(mx, o);
        mx = t > mx ? t : mx;
      }
      if (lane == 0) {
        hinfo->kept = kbase; hinfo->edges = (unsigned long long)total_edges;
        hinfo->rounds = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&ctr->rounds), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hinfo->cand_total = v; hinfo->cand_worst = mx;
        hinfo->heavy_units = 0u; hinfo->light_units = 0u;
        hinfo->route = 1u; hinfo->unit_worst = 0u;
      }
    }
  }
}

// Workgroups per edge shard of a large call.  The resolve's workgroups wait for each other's atomics: ALL of them must be
// resident together, or the resident ones poll for ever on edges whose owners never start.  So the grid is bounded by half
// of what the device keeps resident of this kernel (occupancy x CUs; half: a second call on another stream still fits) --
// 4 on an MI355X (256 CUs x 2 workgroups of 1024 threads), 1 on a device with 64 CUs.
static int resolve_parts() {
  static int cap[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 1;
  if (cap[dev] == 0) {
    int cus = 0, nb = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nms_resolve_kernel<false>, kResolveThreads, 0) != hipSuccess) nb = 0;
    const long long fit = (long long)cus * nb / (2 * kShards);
    cap[dev] = (int)(fit < 1 ? 1 : (fit > 16 ? 16 : fit));
  }
  const int v = (int)tuning(OBB_TUNE_NMS_RESOLVE_PARTS, kResolveParts);
  return v < 1 ? 1 : (v > cap[dev] ? cap[dev] : v);
}

// ---- 9: compaction --------------------------------------------------------------
__device__ __forceinline__ bool kept_flag(const int *__restrict__ cnt, int64_t i) {
  return word_kept(cnt[i]);
}

// One pass, no second launch: a block takes a TICKET (its rank in start order, so every
// block it will wait for is already running -- HIP promises nothing about dispatch order),
// publishes the number of kept boxes of its 1024-box slice with a VALID bit in one agent-scope
// word, and sums the words of all earlier tickets, polling the few that are not there yet.
// The payload IS the flagged word, so no fence is involved; nobody waits for a later ticket.

// PER slices of 1024 boxes per block: the ticket is a returning atomic on ONE word (~90 per
// microsecond), 14 us of tickets alone at 1.25e6 boxes with one slice per block.
template <int PER>
__global__ __launch_bounds__(1024) void nms_compact_kernel(const int *__restrict__ cnt, int64_t n,
                                                           unsigned int *__restrict__ blockcnt,   // [0] ticket, [1 + t] count | VALID
                                                           int64_t *__restrict__ keep,
                                                           NmsCounters *__restrict__ ctr,
                                                           const unsigned long long *__restrict__ ccnts,
                                                           const unsigned int *__restrict__ ucnts,
                                                           NmsHostInfo *__restrict__ hinfo,
                                                           const NmsGridCtr *__restrict__ gctr) {
  __shared__ int wsum[PER][16];
  __shared__ long long red[16];
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = __hip_atomic_fetch_add(&blockcnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned int t = s_ticket;
  const int64_t i0 = (int64_t)t * (PER * 1024) + tid;
  bool f[PER];
  uint64_t mk[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int64_t i = i0 + u * 1024;
    f[u] = (i < n) && kept_flag(cnt, i);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    mk[u] = __ballot(f[u]);
    if (lane == 0) wsum[u][wave] = __popcll(mk[u]);
  }
  __syncthreads();
  int off[PER], tot = 0;   // start of this wave's kept boxes of slice u inside the block
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    off[u] = tot;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int v = wsum[u][k];
      off[u] += (k < wave) ? v : 0;
      tot += v;
    }
  }
  if (tid == 0) __hip_atomic_store(&blockcnt[1 + t], (unsigned int)tot | kCountValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long part = 0;
  for (unsigned int b = tid; b < t; b += 1024) {
    unsigned int v;
    do { v = __hip_atomic_load(&blockcnt[1 + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v & kCountValid));
    part += (long long)(v & ~kCountValid);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  long long sbase = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sbase += red[k];
#pragma unroll
  for (int u = 0; u < PER; ++u)
    if (f[u]) keep[sbase + off[u] + __popcll(mk[u] & ((1ull << lane) - 1ull))] = i0 + u * 1024;
  if (t == gridDim.x - 1 && wave == 0) {  // fold the padded shard counters for the host
    unsigned long long v = ccnts[(size_t)lane * kCandStride], mx = v;
    const unsigned long long ev = ccnts[(size_t)(kShards + lane) * kCandStride];
    mx = ev > mx ? ev : mx;   // fullest segment of either kind
    v += ccnts[(size_t)(2 * kShards + lane) * kCandStride];   // (clipped in line: counted, never listed)
    unsigned int uh = ucnts[lane * kUnitStride], ul = ucnts[(kShards + lane) * kUnitStride], uw = uh + ul;
    for (int o = 32; o > 0; o >>= 1) {
      v += __shfl_xor(v, o);
      const unsigned long long tt = __shfl_xor(mx, o);
      mx = tt > mx ? tt : mx;
      uh += __shfl_xor(uh, o); ul += __shfl_xor(ul, o);
      const unsigned int tw = __shfl_xor(uw, o);
      uw = tw > uw ? tw : uw;
    }
    if (lane == 0) {
      hinfo->kept = sbase + tot; hinfo->edges = ctr->edges; hinfo->rounds = ctr->rounds;
      hinfo->cand_total = v; hinfo->cand_worst = mx;
      hinfo->heavy_units = uh; hinfo->light_units = ul;
      const unsigned int gf = gctr ? gctr->grid_fail : 0u;
      hinfo->route = gf ? (3u | (gf << 8)) : 2u; hinfo->unit_worst = uw;   // 3: the grid route gave up -- repeat without it
    }
  }
}

// ---- fused path for small n: the whole NMS in ONE launch of ONE workgroup ------------
// Detection heads call NMS thousands of times on a few hundred to ~2000 boxes
// (RPN levels, per-image heads; rpn_head.py:92, bbox_nms.py:60-64): there the ten
// launches of the general pipeline cost more than the work.  Same arithmetic, same
// pruning, same edge/resolve formulation; state lives in LDS, records and edges in
// the caller's workspace.
constexpr int kSmallMax = 1024;          // one box per thread
constexpr int kFusedDefault = 128;       // largest n routed to the fused kernel (the direct path is faster above)
// largest n whose sorted scan clips in line.  tools/exp_inline_clip.py: 20k / 50k sparse boxes 78 -> 73 / 91 -> 83 us and
// no loss on dense scenes; from 100k on the 4 waves/SIMD of the bigger kernel cost more (dense 155 -> 171 us) than
// the launch saves (sparse 111 -> 106 us)
constexpr int64_t kInlineDefault = 65536;
constexpr int kDirectDefault = 10240;    // largest n routed to the direct (unsorted) scan (measured crossover 11-12k)
constexpr int kDirectMax = 65536;

template <int KIND, bool F64 = false>
__global__ __launch_bounds__(1024) void nms_small_kernel(const NmsSource src, int n,
                                                         float thr_global, int flags,
                                                         QuadRec *__restrict__ rec,
                                                         uint2 *__restrict__ edges,
                                                         unsigned int ecap,
                                                         int64_t *__restrict__ keep,
                                                         NmsHostInfo *__restrict__ hinfo) {
  constexpr int S = KIND == 0 ? 6 : (KIND == 1 ? 9 : 5);
  __shared__ uint2 qiv[kSmallMax];            // packed (lo, hi|G)
  __shared__ float area[kSmallMax];
  __shared__ uint32_t skey[kSmallMax];
  __shared__ int cnt[kSmallMax];
  __shared__ uint8_t state[kSmallMax];
  __shared__ uint2 queue[16][kNmsQueue];
  __shared__ float red[16][4];
  __shared__ float frame[6];                  // base x,y, scale x,y, cell px x,y
  __shared__ unsigned int n_edges;
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- a: records, intervals, frame
  const float big = 3.0e38f;
  float xl = big, xh = -big, yl = big, yh = -big;
  bool fin = false;
  if (tid == 0) n_edges = 0;
  if (tid < n) {
    float rowf[5];
    double d64[4];
    uint64_t key64 = 0;
    bool wf64 = true;
    const float *row;
    if constexpr (F64) { f64_row(src, tid, rowf, d64, key64, wf64); row = rowf; }
    else row = src.dets + (int64_t)tid * src.pitch;
    Quad q;
    load_quad<KIND>(row, q);
    QuadRec r;
    r.ox = q.ox; r.oy = q.oy; r.ex = q.ex; r.ey = q.ey;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.rx[k] = q.rx[k]; r.ry[k] = q.ry[k]; }
    r.area = q.area; r.skey = score_key(row[S - 1]);
    r.group = src_group<S>(src, tid, row);
    r.thr = src_thr<S>(src, tid, row, thr_global);
    bool wellformed = true;
    if (KIND == 2) hbb_record(row, r, wellformed);
    if constexpr (F64) { rec_put_doubles(r, d64); r.skey = (uint32_t)(key64 >> 32); src.klo[tid] = (uint32_t)key64; wellformed = wf64; }
    rec[tid] = r;
    skey[tid] = r.skey;
    area[tid] = q.area;
    nms_interval(q.ox, q.ex, xl, xh);
    nms_interval(q.oy, q.ey, yl, yh);
    fin = wellformed && (fabs_sel(xl) < big) && (fabs_sel(xh) < big) && (fabs_sel(yl) < big) && (fabs_sel(yh) < big);
  }
  cnt[tid] = 0;
  state[tid] = kUndecided;
  float mnx = fin ? xl : big, mxx = fin ? xh : -big, mny = fin ? yl : big, mxy = fin ? yh : -big;
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min_sel(mnx, __shfl_xor(mnx, o)); mxx = max_sel(mxx, __shfl_xor(mxx, o));
    mny = min_sel(mny, __shfl_xor(mny, o)); mxy = max_sel(mxy, __shfl_xor(mxy, o));
  }
  if (lane == 0) { red[wave][0] = mnx; red[wave][1] = mxx; red[wave][2] = mny; red[wave][3] = mxy; }
  __syncthreads();
  if (tid == 0) {
    float x0 = red[0][0], x1 = red[0][1], y0 = red[0][2], y1 = red[0][3];
    for (int k = 1; k < 16; ++k) {
      x0 = min_sel(x0, red[k][0]); x1 = max_sel(x1, red[k][1]);
      y0 = min_sel(y0, red[k][2]); y1 = max_sel(y1, red[k][3]);
    }
    const float rx = x1 - x0, ry = y1 - y0;
    frame[0] = x0; frame[1] = y0;
    frame[2] = rx > 0.0f ? (float)kQMax / rx : 0.0f;
    frame[3] = ry > 0.0f ? (float)kQMax / ry : 0.0f;
    frame[4] = rx > 0.0f ? (rx / (float)kQMax) * 1.000001f : 0.0f;
    frame[5] = ry > 0.0f ? (ry / (float)kQMax) * 1.000001f : 0.0f;
  }
  __syncthreads();
  {
    auto qz = [](float x, float b, float s) {
      float u = floorf((x - b) * s);
      u = u < 0.0f ? 0.0f : (u > (float)kQMax ? (float)kQMax : u);
      return (uint32_t)u;
    };
    uint2 o = make_uint2(kSentLo, kSentHi);
    if (fin) {
      o.x = qz(xl, frame[0], frame[2]) | (qz(yl, frame[1], frame[3]) << 16);
      o.y = (qz(xh, frame[0], frame[2]) | (qz(yh, frame[1], frame[3]) << 16)) | kGuard;
    }
    qiv[tid] = o;   // threads >= n hold the sentinel
  }
  __syncthreads();

  // ---- b: pair scan (rows r, columns c > r), candidates -> clip -> edges
  Prune pr;
  pr.thr = src_thresholds(src) ? 0.0f : thr_global; pr.cpx = frame[4]; pr.cpy = frame[5];
  uint2 *q = queue[wave];
  int qn = 0;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t *klo_p = src.klo;
  auto drain = [&](bool flush) {
    while (qn >= 64 || (flush && qn > 0)) {
      const int take = qn < 64 ? qn : 64;
      qn -= take;
      __builtin_amdgcn_wave_barrier();
      bool hit = false;
      uint32_t src = 0, dst = 0;
      if (lane < take) {
        const uint2 prr = q[qn + lane];
        Quad A, B;
        uint32_t ka, kb;
        int ga, gb;
        float ta, tb;
        load_rec(rec, prr.x, A, ka, ga, ta);
        load_rec(rec, prr.y, B, kb, gb, tb);
        bool b_first = (kb < ka) || (kb == ka && prr.y < prr.x);
        if constexpr (F64) {
          if (kb == ka) {      // (the low words were written by this block before the first barrier)
            const uint32_t la = klo_p[prr.x], lb = klo_p[prr.y];
            b_first = (lb < la) || (lb == la && prr.y < prr.x);
          }
        }
        src = b_first ? prr.y : prr.x;
        dst = b_first ? prr.x : prr.y;
        swap_if(b_first, A.ox, B.ox); swap_if(b_first, A.oy, B.oy);
        swap_if(b_first, A.ex, B.ex); swap_if(b_first, A.ey, B.ey);
        swap_if(b_first, A.area, B.area);
#pragma unroll
        for (int k = 0; k < 4; ++k) { swap_if(b_first, A.rx[k], B.rx[k]); swap_if(b_first, A.ry[k], B.ry[k]); }
        hit = (ga == gb) && narrow_hit(A, B, b_first ? tb : ta, flags);
      }
      const uint64_t mk = __ballot(hit);
      if (mk) {
        const int leader = __ffsll((long long)mk) - 1;
        unsigned int pos0 = 0;
        if (lane == leader) pos0 = atomicAdd(&n_edges, (unsigned int)__popcll(mk));
        pos0 = (unsigned int)__shfl((int)pos0, leader);
        if (hit) {
          const unsigned int pos = pos0 + __popcll(mk & lt_mask);
          if (pos < ecap) edges[pos] = make_uint2(src, dst);
          atomicAdd(&cnt[dst], 1);
        }
      }
    }
  };
  const int nchunks = (n + kTile - 1) / kTile;          // column chunks of 256
  for (int cb = 0; cb < nchunks; ++cb) {
    uint2 col[4];
    float car[4];
    const int c0 = cb * kTile + lane * kNmsColsPerLane;
#pragma unroll
    for (int k = 0; k < 4; ++k) { col[k] = qiv[c0 + k]; car[k] = area[c0 + k]; }
    const int rend = min(n, cb * kTile + kTile - 1);    // rows that have a column > r in this chunk
    for (int r = wave; r < rend; r += 16) {
      const uint2 rv = qiv[r];                           // LDS broadcast
      const uint32_t t0 = qdisjoint(rv.x, rv.y, col[0].x, col[0].y);
      const uint32_t t1 = qdisjoint(rv.x, rv.y, col[1].x, col[1].y);
      const uint32_t t2 = qdisjoint(rv.x, rv.y, col[2].x, col[2].y);
      const uint32_t t3 = qdisjoint(rv.x, rv.y, col[3].x, col[3].y);
      const uint32_t tt[4] = {t0, t1, t2, t3};
      bool any_c = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) any_c = any_c || (tt[k] == 0u && c0 + k > r);
      if (!__any(any_c)) continue;
      const float ra = area[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool cnd = (tt[k] == 0u) && (c0 + k > r);
        if (pr.thr > 0.0f && __any(cnd)) {
          cnd = cnd && prune_area(pr, ra, car[k]);
          if (__any(cnd)) cnd = cnd && prune_aabb(pr, rv.x, rv.y, ra, col[k].x, col[k].y, car[k]);
        }
        const uint64_t mk = __ballot(cnd);
        if (mk) {
          if (cnd) q[qn + __popcll(mk & lt_mask)] = make_uint2((uint32_t)r, (uint32_t)(c0 + k));
          qn += __popcll(mk);
        }
      }
      drain(false);
    }
  }
  drain(true);
  __syncthreads();

  // ---- c: resolve (state / cnt in LDS, edges in the workspace)
  const unsigned int found = n_edges;
  long long live = found <= ecap ? (long long)found : 0;
  long long rounds = 0;
  int stalled = 0;
  while (live > 0) {
    long long w = 0;
    for (long long base = 0; base < live; base += 1024) {
      const long long e = base + tid;
      bool kp = false;
      uint2 ed = make_uint2(0u, 0u);
      if (e < live) {
        ed = edges[e];
        if (state[ed.y] == kUndecided) {
          uint8_t si = state[ed.x];
          if (si == kUndecided && cnt[ed.x] == 0) si = kKept;
          if (si == kKept) {
            state[ed.y] = kSuppressed;
          } else if (si == kSuppressed) {
            if (atomicSub(&cnt[ed.y], 1) == 1) state[ed.y] = kKept;
          } else {
            kp = true;
          }
        }
      }
      const uint64_t mk = __ballot(kp);
      if (lane == 0) wsum[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int v = wsum[k];
        off += (k < wave) ? v : 0;
        tot += v;
      }
      if (kp) edges[w + off + __popcll(mk & lt_mask)] = ed;
      w += tot;
      __syncthreads();
    }
    stalled = (w == live) ? stalled + 1 : 0;
    live = w;
    ++rounds;
    __syncthreads();
    if (stalled >= 2) break;
  }

  // ---- d: ascending keep indices + counters
  const bool f = (tid < n) && ((state[tid] == kKept) || (state[tid] == kUndecided && cnt[tid] == 0));
  const uint64_t mk = __ballot(f);
  if (lane == 0) wsum[wave] = __popcll(mk);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int v = wsum[k];
    off += (k < wave) ? v : 0;
    tot += v;
  }
  if (f) keep[off + __popcll(mk & lt_mask)] = tid;
  if (tid == 0) {
    hinfo->edges = found;
    hinfo->kept = tot;
    hinfo->rounds = rounds;
    hinfo->cand_total = found;
    hinfo->cand_worst = 0ull;
    hinfo->heavy_units = 0u; hinfo->light_units = 0u;
    hinfo->route = 0u; hinfo->unit_worst = 0u;
  }
}

// ---- horizontal boxes, `>=` rule, threshold <= 0 (or NaN) ------------------------------------------------------------
// nms_cpu.cpp:35-58 as it stands: for i in score order, not suppressed: every later j with ovr(i, j) >= thr is suppressed
// -- with thr <= 0 that is EVERY later box of the group whose ovr is not negative or NaN (inverted boxes), overlapping or
// not, so the broad phase of the chain above (which pairs only boxes whose intervals meet) does not apply.  One
// workgroup; a thread owns the boxes t, t + 1024, ...; a round = the best live box i (64-bit key of the score, then the
// index: the order of every other route) is kept, then every owner tests its live boxes against i and names its best
// survivor for the next round.  Rounds = kept boxes (one per group for well-formed boxes).  A NaN threshold suppresses
// nothing (`ovr >= NaN` is false): every box is kept.
template <bool F64>
__global__ __launch_bounds__(1024) void nms_hbb_allpairs_kernel(const NmsSource src, int64_t n, float thr, uint32_t *__restrict__ state,
                                                                int64_t *__restrict__ keep, NmsHostInfo *__restrict__ hinfo) {
  __shared__ unsigned long long wkey[16];
  __shared__ long long widx[16];
  __shared__ long long s_win;
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool all = thr != thr;
  for (int64_t j = tid; j < n; j += 1024) state[j] = all ? 2u : 0u;     // 0 live, 1 suppressed, 2 kept (owner-private words)
  auto key_of = [&](int64_t j) -> unsigned long long {
    if constexpr (F64) return score_key64(src.dets64[j * src.pitch + 4]);
    else return (unsigned long long)score_key(src.dets[j * src.pitch + 4]) << 32;
  };
  auto group_of = [&](int64_t j) -> int32_t {
    if constexpr (F64) return src.group ? src.group[j] : 0;
    else return src_group<5>(src, j, src.dets + j * src.pitch);
  };
  long long rounds = 0;
  long long win = -1;
  while (!all) {
    // the winner's row (block-uniform)
    double wd[4] = {0, 0, 0, 0};
    float wf[4] = {0, 0, 0, 0}, warea = 0.0f;
    int32_t wg = 0;
    if (win >= 0) {
      if constexpr (F64) { for (int k = 0; k < 4; ++k) wd[k] = src.dets64[win * src.pitch + k]; }
      else { for (int k = 0; k < 4; ++k) wf[k] = src.dets[win * src.pitch + k]; warea = hbb_area(wf[0], wf[1], wf[2], wf[3]); }
      wg = group_of(win);
    }
    unsigned long long bk = ~0ull;
    long long bi = -1;
    for (int64_t j = tid; j < n; j += 1024) {
      if (state[j] != 0u) continue;
      if (win >= 0 && group_of(j) == wg) {
        bool hit;
        if constexpr (F64) {
          const double *r = src.dets64 + j * src.pitch;
          const double aarea = ((wd[2] - wd[0]) + 1.0) * ((wd[3] - wd[1]) + 1.0), barea = ((r[2] - r[0]) + 1.0) * ((r[3] - r[1]) + 1.0);
          const double xx1 = dmax_sel(wd[0], r[0]), yy1 = dmax_sel(wd[1], r[1]);
          const double xx2 = dmin_sel(wd[2], r[2]), yy2 = dmin_sel(wd[3], r[3]);
          const double w = dmax_sel(0.0, (xx2 - xx1) + 1.0), h = dmax_sel(0.0, (yy2 - yy1) + 1.0);
          const double inter = w * h;
          hit = inter / ((aarea + barea) - inter) >= (double)thr;
        } else {
          const float *r = src.dets + j * src.pitch;
          const float b[4] = {r[0], r[1], r[2], r[3]};
          hit = hbb_overlap(wf, warea, b, hbb_area(b[0], b[1], b[2], b[3])) >= thr;
        }
        if (hit) { state[j] = 1u; continue; }
      }
      const unsigned long long kj = key_of(j);
      if (bi < 0 || kj < bk) { bk = kj; bi = j; }          // (ascending j per thread: the first of equal keys stays)
    }
    // block-wide minimum of (key, index)
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long ok = __shfl_xor(bk, o);
      const long long oi = __shfl_xor(bi, o);
      if (oi >= 0 && (bi < 0 || ok < bk || (ok == bk && oi < bi))) { bk = ok; bi = oi; }
    }
    if (lane == 0) { wkey[wave] = bk; widx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      unsigned long long k0 = wkey[0];
      long long i0 = widx[0];
      for (int w2 = 1; w2 < 16; ++w2) {
        const unsigned long long ok = wkey[w2];
        const long long oi = widx[w2];
        if (oi >= 0 && (i0 < 0 || ok < k0 || (ok == k0 && oi < i0))) { k0 = ok; i0 = oi; }
      }
      s_win = i0;
    }
    __syncthreads();
    win = s_win;
    __syncthreads();
    if (win < 0) break;
    if ((win & 1023) == tid) state[win] = 2u;             // (its owner)
    ++rounds;
  }
  __syncthreads();
  // ascending keep indices
  long long total = 0;
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t j = base + tid;
    const bool f = j < n && state[j] == 2u;
    const uint64_t mk = __ballot(f);
    if (lane == 0) wsum[wave] = __popcll(mk);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int v = wsum[k]; off += (k < wave) ? v : 0; tot += v; }
    if (f) keep[total + off + __popcll(mk & ((1ull << lane) - 1ull))] = j;
    total += tot;
    __syncthreads();
  }
  if (tid == 0) {
    hinfo->edges = 0ull; hinfo->kept = total; hinfo->rounds = rounds;
    hinfo->cand_total = 0ull; hinfo->cand_worst = 0ull;
    hinfo->heavy_units = 0u; hinfo->light_units = 0u;
    hinfo->route = 0u; hinfo->unit_worst = 0u;
  }
}

// ---- optional per-phase timing (HIP events on the caller's stream) ----------------
constexpr int kPhases = OBB_NMS_PHASES;
struct NmsProfile {
  int level = 0;  // 0 off, 1 scan kernel only (two events), 2 every phase
  bool on = false;
  bool have = false;
  hipEvent_t ev[kPhases + 1] = {};
  bool created = false;
  float ms[kPhases] = {};
};
static thread_local NmsProfile g_prof;

// ---- workspace layout ------------------------------------------------------------
struct NmsLayout {
  size_t rec, iv, ar, qb, ckey, crank, gsrc, tsrc, gtab, qs, pf, pg, pt, tgr, sgr, frame, tbox, sbox, zero_begin, cnt, ctr, ccnts, ucnts, super, hist, scanflags, ticket,
      zero_end, blockcnt, cands, edges, total;
  unsigned long long cap;         // candidate pairs == overlap edges capacity
  unsigned long long unit_worst;  // kShards x the most units one shard of the unit list can get
  int64_t ntiles;
  int grid_log2;                  // grid route: cells per axis = 2^grid_log2 at most (0: no grid table in this layout)
  size_t gover;                   // grid route: the overflow list (entries that found their cell full)
  size_t gbig;                    // grid route: the boxes over too many cells
  int grid_bits;                  // ... and 2^grid_bits cells in the table (>= 2 * grid_log2: the groups of a grouped call share it)
  size_t retain, gcnt;            // round 6: the retained-state header (offset 0) and the grid route's cell counters, at offsets that
                                  // depend on grid_bits only -- what a call leaves there for the next one survives a change of n
};

// grid route: cells per axis at most (2^g) and cells of the table (2^bits) for a call of n boxes; false: no grid for this n
static bool grid_geometry(int64_t n, int &g, int &bits) {
  g = 0; bits = 0;
  if (!(n >= 256 && n <= (1 << 24))) return false;
  g = 6;
  while (g < kGridMaxLog2 && (1ll << (2 * g)) * 2 < n) ++g;      // (x 8 instead of x 2 -- a quarter of the probe's waves -- measured no faster: the cell size follows the extents, not this bound)
  // table: the spatial grid, or -- grouped calls: the groups' grids share the table -- about 2.5 boxes per cell: with
  // ~1.8 entries per box that is 4.5 entries per cell on average and a 33rd entry (the route gives up) a 6-sigma
  // event of the compound-Poisson count (2^18 cells for 1.25 M boxes in 24 000 groups: 9.7 on average, dozens of
  // overflowing cells in every call)
  bits = 2 * g;                                               // (half that table for 1.25 M grouped boxes: the same times)
  while (bits < kGridMaxBits && ((int64_t)5 << bits) < 2 * n) ++bits;
  return true;
}

static int nms_layout(int64_t n, int64_t edge_capacity, NmsLayout &L) {
  if (edge_capacity <= 0) {
    edge_capacity = 64 * n;
    if (edge_capacity < (1 << 18)) edge_capacity = 1 << 18;
  }
  L.cap = (unsigned long long)edge_capacity;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  const size_t nt = (nn + kTile - 1) / kTile;
  L.ntiles = (int64_t)nt;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  L.grid_log2 = 0; L.grid_bits = 0;
  L.retain = take(sizeof(NmsRetain));
  L.gcnt = off;
  {
    int g = 0, bits = 0;
    if (grid_geometry(n, g, bits)) {
      L.grid_log2 = g; L.grid_bits = bits;
      // (table and cell counters on 64-KiB boundaries: the insert kernel -- scattered atomics and 16-B stores -- is 2.5 us
      // slower at 100 000 - 131 072 boxes with either of them at certain other offsets, whatever the code; DESIGN_LOG)
      off = align_up(off, 65536);
      L.gcnt = take(((size_t)1 << bits) * sizeof(uint32_t));
      off = align_up(off, 65536);
    }
  }
  L.rec = take(nn * sizeof(QuadRec));
  L.iv = take(nn * sizeof(float4));       // (direct route only)
  L.ar = take(nn * sizeof(float));
  L.qb = take(nn * sizeof(uint2));
  L.ckey = take(nn * 4);
  L.crank = take(nn * 4);
  L.gsrc = take(nn * 4);       // per-box group / threshold derived by the prep kernel (NmsSource modes 1, 2)
  L.tsrc = take(nn * 4);
  // grid route: (2^g x 2^g cells) x kGridCap entries of 16 bytes; g from n (about half a cell per box), sizes where the route applies
  L.gtab = off;
  if (L.grid_log2 > 0) {
    off = align_up(off, 65536);
    L.gtab = take(((size_t)1 << L.grid_bits) * kGridCap * sizeof(uint4));
    L.gover = take((size_t)kGridOverMax * (sizeof(uint4) + sizeof(uint32_t)) + (size_t)kGridBigMax * sizeof(uint32_t));     // overflow entries, then their cells, then the big list
    L.gbig = L.gover + (size_t)kGridOverMax * (sizeof(uint4) + sizeof(uint32_t));
  }
  L.qs = take((nt * kTile + kRowChunk) * sizeof(uint2));
  L.pf = take((nt * kTile) * sizeof(uint4));
  L.pg = take((nt * kTile) * sizeof(int32_t));
  L.pt = take((nt * kTile) * sizeof(float));
  L.sgr = take(nt * kSub * sizeof(uint32_t));
  L.frame = take(sizeof(NmsFrame));
  L.sbox = take(nt * kSub * sizeof(uint2));
  {  // shard s takes tiles s, s + 64, ...; tile A lists at most (nt - 1 - A) * kSub units (its pairs with later tiles:
     // nms_cull_kernel) or A * kSub (with earlier tiles: nms_cullbox_kernel) -- the bound below covers both
    const unsigned long long m = (nt + kShards - 1) / kShards;
    L.unit_worst = (m * nt - kShards * (m * (m - 1) / 2)) * kSub * kShards;
  }
  L.zero_begin = off;
  L.cnt = take(nn * 4);
  L.ctr = take(sizeof(NmsCounters));
  L.ccnts = take((size_t)3 * kShards * kCandStride * sizeof(unsigned long long));  // candidates | edges | candidates clipped in line
  L.ucnts = take((size_t)2 * kShards * kUnitStride * sizeof(unsigned int));        // heavy | light scan units
  L.super = take(((nt + kSuper - 1) / kSuper) * 8 * 4);   // supertile boxes / bucket ranges (maxima: zero = empty)
  // ticket + one flagged count per 1024 boxes (nms_compact_kernel) -- or per SHARD (the compaction fused into
  // nms_resolve_kernel publishes blockcnt[1 + shard], shard < kShards, whatever n is): sized for both users
  // (with 64 words, shard 63 wrote the first word of `scanflags` for 10 240 < n <= 64 512)
  L.blockcnt = take((((nn + 1023) / 1024 > (size_t)kShards ? (nn + 1023) / 1024 : (size_t)kShards) + 1) * 4);
  L.scanflags = take((kCells / kScanBlockCells + 1) * 4);   // multi-block cell scan: ticket + one flagged total per block
  L.ticket = take(4);                                // nms_cullbox_kernel: tile tickets
  L.tbox = take(nt * sizeof(uint2));                 // (zero = "not published yet" for nms_cullbox_kernel)
  L.tgr = take(nt * sizeof(uint32_t));
  L.hist = take(((size_t)1 << 18) * 4);   // (kCells; LAST of the zeroed range: only the cells a call uses are cleared.  The grid route's cell counters lived here until round 6: L.gcnt)
  static_assert(kCells == (1 << 18), "hist");
  L.zero_end = off;
  // variable part, all 8-byte entries: [cands | edges | units]; candidates and edges
  // share one capacity, the unit list gets 1/8 of it (never more than unit_worst).
  // The split is re-derived from the actual workspace size at run time (nms_split).
  L.cands = off;
  L.edges = off;
  unsigned long long ucap = L.cap / 8 + 4096;
  if (ucap > L.unit_worst) ucap = L.unit_worst;
  off += (2 * (size_t)L.cap + (size_t)ucap) * sizeof(uint2);
  L.total = off;
  return OBB_OK;
}

// entries (8 bytes each) of the variable part -> per-shard pair capacity + unit capacity
static void nms_split(unsigned long long entries, unsigned long long unit_worst,
                      unsigned long long &segcap, unsigned long long &ucap) {
  // inverse of nms_layout: entries = 2 cap + min(cap / 8 + 4096, unit_worst)
  if (entries >= unit_worst && (entries - unit_worst) / 16 + 4096 >= unit_worst) {
    ucap = unit_worst;
  } else {
    const unsigned long long cap = entries > 4096 ? (entries - 4096) * 8 / 17 : 0;
    ucap = entries - 2 * cap;
    if (ucap > unit_worst) ucap = unit_worst;
  }
  if (ucap > 0xFFFFFFFFull) ucap = 0xFFFFFFFFull;
  ucap -= ucap % kShards;   // kShards equal segments
  segcap = (entries - ucap) / 2 / kShards;
}

// ---- capacity hints (round 4: moved here from the Python wrapper, so that C-ABI callers get them too) ---------------
// A call whose pair lists overflow the default capacity (64 n pairs) is repeated with the capacity the library names --
// twice the time.  A caller that does this every time (the RPN of a dense tile: 8768 boxes, 5.7e5 overlapping pairs)
// should start there: the calling THREAD remembers, per (kind, size class of n), the capacity its last overflowing call
// needed; obb_nms_workspace_bytes(n, kind, 0) sizes the default workspace with it, and a call that used less than an
// eighth of a hinted capacity drops the hint again.  Thread-local, no locks; forgetting a hint costs one repeated call.
struct CapHint { int kind, bits; unsigned long long cap; };
constexpr int kCapHints = 16;
static thread_local CapHint g_cap_hints[kCapHints] = {};
static int size_class(int64_t n) { int b = 0; while (n > 0) { ++b; n >>= 1; } return b; }
static unsigned long long cap_hint_get(int64_t n, int kind) {
  const int bits = size_class(n);
  for (const CapHint &h : g_cap_hints) if (h.cap && h.kind == kind && h.bits == bits) return h.cap;
  return 0ull;
}
static void cap_hint_put(int64_t n, int kind, unsigned long long cap) {
  const int bits = size_class(n);
  CapHint *slot = nullptr;
  for (CapHint &h : g_cap_hints) if (h.cap && h.kind == kind && h.bits == bits) { slot = &h; break; }
  if (!slot) for (CapHint &h : g_cap_hints) if (!h.cap) { slot = &h; break; }
  if (!slot) slot = &g_cap_hints[(unsigned)(bits + 7 * kind) % kCapHints];     // table full: replace one
  slot->kind = kind; slot->bits = bits; slot->cap = cap;
}
// after a call's verdict: rc == OBB_EWORKSPACE -> remember `need`; a fitting call that used `used` pairs of a hinted capacity
static void cap_hint_note(int64_t n, int kind, int rc, unsigned long long need, unsigned long long used) {
  if (rc == OBB_EWORKSPACE) { if (need) cap_hint_put(n, kind, need); return; }
  const unsigned long long h = cap_hint_get(n, kind);
  if (rc == OBB_OK && h && 8ull * used < h) cap_hint_put(n, kind, 0ull);
}

// grid route: after a failed try (a crowded cell) the thread's next kGridSkipCalls calls of the size class go straight to the
// sorted route
static thread_local int g_last_route = -1;     // obb_nms_last_route(): 0 fused, 1 direct, 2 sorted, 4 grid, 5 grid gave up -> sorted
struct GridSkip { int kind, bits, left; };
static thread_local GridSkip g_grid_skip[8] = {};
static bool grid_skip_take(int64_t n, int kind) {
  const int bits = size_class(n);
  for (GridSkip &h : g_grid_skip)
    if (h.left > 0 && h.kind == kind && h.bits == bits) { --h.left; return true; }
  return false;
}
void nms_grid_hints_clear() { for (GridSkip &h : g_grid_skip) h.left = 0; }
static void grid_skip_put(int64_t n, int kind) {
  const int bits = size_class(n);
  GridSkip *slot = &g_grid_skip[(unsigned)(bits + 3 * kind) % 8];
  for (GridSkip &h : g_grid_skip) if (h.left <= 0) { slot = &h; break; }
  slot->kind = kind; slot->bits = bits; slot->left = kGridSkipCalls;
}

}  // namespace obb

extern "C" size_t obb_nms_workspace_bytes(int64_t n, int kind, int64_t edge_capacity) {
  obb::NmsLayout L;
  if (n < 0) return 0;
  if (edge_capacity <= 0) {        // the default capacity, or what this thread's last overflowing call of the size class needed
    const unsigned long long h = obb::cap_hint_get(n, kind);
    long long def = 64 * n;
    if (def < (1 << 18)) def = 1 << 18;
    edge_capacity = h > (unsigned long long)def ? (int64_t)h : 0;
  }
  if (obb::nms_layout(n, edge_capacity, L) != OBB_OK) return 0;
  return L.total;
}

namespace obb {

// (NmsEnqueued -- the host-side state of one enqueued call: what the capacity check needs afterwards -- lives in
// obb_nms_internal.h)

// capacities the kernels were (or would be) launched with for a workspace of this size
static int nms_capacities(int64_t n, size_t workspace_bytes, NmsLayout &L, NmsEnqueued &q) {
  int rc = nms_layout(n, 1, L);
  if (rc != OBB_OK) return rc;
  const size_t fixed = L.cands;
  unsigned long long segcap = 0, ucap64 = 0;
  if (workspace_bytes > fixed) nms_split((workspace_bytes - fixed) / sizeof(uint2), L.unit_worst, segcap, ucap64);
  if (segcap == 0 || ucap64 == 0)
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: workspace %zu bytes < minimum %zu", workspace_bytes, L.total + 2 * kShards * sizeof(uint2));
  q.segcap = segcap;
  q.cap = segcap * kShards;
  q.ucap = (unsigned int)ucap64;
  if (q.ucap < kShards)
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: workspace %zu bytes < minimum %zu", workspace_bytes, L.total + 2 * kShards * sizeof(uint2));
  return OBB_OK;
}

// The verdict on a finished call from the record its last kernel wrote (the capacity rules: nms_need_capacity,
// obb_nms_internal.h -- shared with the device-side check of the tiled phases).
static int nms_verdict(const NmsHostInfo &h, const NmsEnqueued &q, int64_t *info_host) {
  if (info_host) {
    info_host[0] = h.kept;
    info_host[1] = (int64_t)h.edges;
    info_host[2] = h.rounds;
    info_host[3] = (int64_t)h.cand_total;
  }
  if (h.rounds == kStalled) {
    if (info_host) info_host[0] = 0;
    return set_err(OBB_ESTALLED, "obb_nms_f32: the resolve kernel stopped without converging (its workgroups must be resident together: "
                   "another persistent kernel on the device?); the keep list is not valid -- repeat the call");
  }
  int why = 0;
  const unsigned long long need = nms_need_capacity(h, q.cap, q.segcap, q.ucap, &why);
  if (need == 0) return OBB_OK;
  if (info_host) { info_host[0] = 0; info_host[3] = (int64_t)need; }
  if (why == 1)
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: %llu overlap edges exceed the pair capacity %llu; retry with edge_capacity >= %llu",
                   h.edges, q.cap, need);
  if (why == 2)
    return set_err(OBB_EWORKSPACE, "obb_nms_f32: %u scan units in one of %d segments exceed the segment capacity %u; retry with edge_capacity >= %llu",
                   h.unit_worst, kShards, q.ucap / kShards, need);
  return set_err(OBB_EWORKSPACE, "obb_nms_f32: %llu candidate pairs / %llu overlap edges exceed the pair capacity %llu; retry with edge_capacity >= %llu",
                 h.cand_total, h.edges, q.cap, need);
}

// argument checks shared by the synchronous and the asynchronous entry points (no HIP call:
// they also run on a machine without a GPU)
static int nms_validate(const char *who, const float *dets_dev, int64_t n, int kind, float iou_thr,
                        bool thr_dev, int flags_in, const int64_t *keep_dev,
                        const void *workspace_dev) {
  if (n < 0) return set_err(OBB_EINVAL, "%s: negative n", who);
  if (kind != OBB_KIND_RBOX && kind != OBB_KIND_POLY && kind != OBB_KIND_HBB) return set_err(OBB_EINVAL, "%s: bad kind %d", who, kind);
  if (flags_in & ~(OBB_NMS_STRICT | OBB_NMS_RETAIN)) return set_err(OBB_EINVAL, "%s: unknown flags 0x%x", who, flags_in);
  // (Horizontal boxes with a threshold <= 0 under the `>=` rule: nms_cpu.cpp:56 then holds for DISJOINT boxes too -- every
  // later box of a group falls to the first one unless its ovr is negative or NaN.  Served exactly, by
  // nms_hbb_allpairs_kernel: nms_enqueue_source.)
  if (n == 0) return OBB_OK;   // nothing is read or written
  if (n > (int64_t)kTile * 65535) return set_err(OBB_EINVAL, "%s: n = %lld exceeds %lld", who, (long long)n, (long long)kTile * 65535);
  if (!dets_dev || !keep_dev || !workspace_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (reinterpret_cast<uintptr_t>(workspace_dev) & 255u) return set_err(OBB_EINVAL, "%s: workspace must be 256-byte aligned", who);
  return OBB_OK;
}

// Validates, lays out the workspace and enqueues every kernel of one call on `stream`; the
// last kernel writes the NmsHostInfo record to `result` (device-visible memory: the pinned
// buffer of the synchronous entry points, or the caller's device record of the async one).
void nms_retain_forget(const void *ws);      // (below: the calling thread's notes of retained workspaces)

int nms_enqueue_source(const char *who, NmsSource src, int64_t n, int kind, float iou_thr, int flags_in,
                       int64_t *keep_dev, void *result_v, void *workspace_dev,
                       size_t workspace_bytes, hipStream_t st, NmsEnqueued &q) {
  NmsHostInfo *result = static_cast<NmsHostInfo *>(result_v);
  // any call outside the retained protocol (another entry point, a tiled phase, a plain or repeated call) may write where a
  // retained workspace keeps its state: the thread's note of it ends here, whatever the entry point
  if (src.retain == 0 && workspace_dev) nms_retain_forget(workspace_dev);
  const bool has_thr = src.mode == 2 ? src.thr_table != nullptr : src.thrs != nullptr;
  int rc = nms_validate(who, src.dets64 ? reinterpret_cast<const float *>(src.dets64) : src.dets, n, kind, iou_thr, has_thr, flags_in, keep_dev, workspace_dev);
  if (rc != OBB_OK) return rc;
  if (!result) return set_err(OBB_EINVAL, "%s: null result record", who);
  {
    const int S = kind == OBB_KIND_RBOX ? 6 : (kind == OBB_KIND_POLY ? 9 : 5);
    if (src.pitch < (src.dets64 ? 5 : S + (src.mode == 2 ? 2 : 0))) return set_err(OBB_EINVAL, "%s: row pitch %lld below the %d columns read", who, (long long)src.pitch, S + (src.mode == 2 ? 2 : 0));
    if (src.mode < 0 || src.mode > 2) return set_err(OBB_EINVAL, "%s: bad source mode", who);
    if (n > 0 && src.mode == 1 && (!src.a64 || !src.b64)) return set_err(OBB_EINVAL, "%s: null label / patch array", who);
  }
  const bool f64 = src.dets64 != nullptr;
  if (f64 && (kind != OBB_KIND_HBB || src.mode != 0)) return set_err(OBB_EINVAL, "%s: float64 rows are horizontal boxes with plain group / threshold arrays", who);
  const int flags = (flags_in & OBB_NMS_STRICT ? kFlagStrict : 0) | (kind == OBB_KIND_HBB ? kFlagHbb : 0) | (f64 ? kFlagF64 : 0);
  if (kind == OBB_KIND_HBB && !(flags_in & OBB_NMS_STRICT) && !has_thr && !(iou_thr > 0.0f)) {
    // nms_cpu.cpp:56 with a threshold <= 0 (or NaN): every pair of a group counts, not only the overlapping ones
    if (workspace_bytes < (size_t)n * sizeof(uint32_t)) return set_err(OBB_EWORKSPACE, "%s: workspace %zu bytes < %lld", who, workspace_bytes, (long long)n * 4);
    q = NmsEnqueued();
    q.fused = true;
    q.cap = 1;
    if (f64) hipLaunchKernelGGL(nms_hbb_allpairs_kernel<true>, dim3(1), dim3(1024), 0, st, src, n, iou_thr, static_cast<uint32_t *>(workspace_dev), keep_dev, result);
    else hipLaunchKernelGGL(nms_hbb_allpairs_kernel<false>, dim3(1), dim3(1024), 0, st, src, n, iou_thr, static_cast<uint32_t *>(workspace_dev), keep_dev, result);
    OBB_LAUNCH_CHECK();
    return OBB_OK;
  }

  // pair capacity = whatever the caller's workspace holds beyond the fixed part,
  // split evenly between the candidate lists and the edge list
  NmsLayout L;
  rc = nms_capacities(n, workspace_bytes, L, q);
  if (rc != OBB_OK) return rc;
  const unsigned long long cap = q.cap, segcap = q.segcap;
  const unsigned int ucap = q.ucap;

  char *ws = static_cast<char *>(workspace_dev);
  QuadRec *rec = reinterpret_cast<QuadRec *>(ws + L.rec);
  float4 *iv = reinterpret_cast<float4 *>(ws + L.iv);
  float *ar = reinterpret_cast<float *>(ws + L.ar);
  uint2 *qb = reinterpret_cast<uint2 *>(ws + L.qb);
  uint32_t *ckey = reinterpret_cast<uint32_t *>(ws + L.ckey);
  uint32_t *crank = reinterpret_cast<uint32_t *>(ws + L.crank);
  uint32_t *hist = reinterpret_cast<uint32_t *>(ws + L.hist);
  uint2 *qs = reinterpret_cast<uint2 *>(ws + L.qs);
  uint4 *pf = reinterpret_cast<uint4 *>(ws + L.pf);
  // per-box group / threshold as the later stages read them: the caller's arrays, or what the prep kernel derives
  const int32_t *group_dev = src.group;
  const float *thr_dev = src.thrs;
  src.gout = nullptr; src.tout = nullptr;
  if (src.mode != 0) {
    src.gout = reinterpret_cast<int32_t *>(ws + L.gsrc);
    group_dev = src.gout;
    thr_dev = nullptr;
    if (src.mode == 2 && src.thr_table) { src.tout = reinterpret_cast<float *>(ws + L.tsrc); thr_dev = src.tout; }
  }
  src.klo = f64 ? reinterpret_cast<uint32_t *>(ws + L.tsrc) : nullptr;     // (mode 0 derives no thresholds: the slot is free)
  const uint32_t *klo = src.klo;
  int32_t *pg = group_dev ? reinterpret_cast<int32_t *>(ws + L.pg) : nullptr;
  float *pt = thr_dev ? reinterpret_cast<float *>(ws + L.pt) : nullptr;
  uint32_t *tgr = reinterpret_cast<uint32_t *>(ws + L.tgr);
  uint32_t *sgr = group_dev ? reinterpret_cast<uint32_t *>(ws + L.sgr) : nullptr;
  NmsFrame *frame = reinterpret_cast<NmsFrame *>(ws + L.frame);
  const float *cellpx = &frame->cpx;       // (cpx, cpy)
  uint2 *tbox = reinterpret_cast<uint2 *>(ws + L.tbox);
  uint2 *sbox = reinterpret_cast<uint2 *>(ws + L.sbox);
  int *cnt = reinterpret_cast<int *>(ws + L.cnt);
  NmsCounters *ctr = reinterpret_cast<NmsCounters *>(ws + L.ctr);
  unsigned long long *ccnts = reinterpret_cast<unsigned long long *>(ws + L.ccnts);
  unsigned int *ucnts = reinterpret_cast<unsigned int *>(ws + L.ucnts);
  unsigned int *super = reinterpret_cast<unsigned int *>(ws + L.super);
  unsigned int *blockcnt = reinterpret_cast<unsigned int *>(ws + L.blockcnt);
  unsigned int *scanflags = reinterpret_cast<unsigned int *>(ws + L.scanflags);
  unsigned int *ticket = reinterpret_cast<unsigned int *>(ws + L.ticket);
  uint2 *cands = reinterpret_cast<uint2 *>(ws + L.cands);
  uint2 *edges = cands + cap;
  uint2 *units = edges + cap;
  const int nt = (int)L.ntiles;

  unsigned pblocks = (unsigned)((n + 255) / 256);
  if (pblocks > (unsigned)kPrepBlocks) pblocks = kPrepBlocks;

  NmsProfile &prof = g_prof;
  const bool timed = prof.on;
  q.timed = timed;
  if (timed && !prof.created) {
    for (int k = 0; k <= kPhases; ++k) OBB_HIP_TRY(hipEventCreate(&prof.ev[k]));
    prof.created = true;
  }
#define OBB_PHASE(k) do { if (timed && (prof.level >= 2 || (k) == 3 || (k) == 4)) OBB_HIP_TRY(hipEventRecord(prof.ev[k], st)); } while (0)
  int64_t fused_max = tuning(OBB_TUNE_NMS_FUSED_MAX, kFusedDefault);
  int64_t direct_max = tuning(OBB_TUNE_NMS_DIRECT_MAX, kDirectDefault);
  if (fused_max > kSmallMax) fused_max = kSmallMax;
  if (direct_max > kDirectMax) direct_max = kDirectMax;
  const bool fused = n <= fused_max;
  const bool direct = !fused && n <= direct_max;
  q.fused = fused; q.direct = direct;
  // sorted route: up to here the scan clips its own candidates (one launch less on a latency-bound chain)
  const bool inline_clip = n <= tuning(OBB_TUNE_NMS_INLINE_MAX, kInlineDefault);
  if (fused) {
    // fused single-launch path; the edge list may use the whole pair capacity
    const unsigned long long ecap64 = 2ull * cap;
    const unsigned int ecap = ecap64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)ecap64;
    if (kind == OBB_KIND_RBOX)
      hipLaunchKernelGGL(nms_small_kernel<0>, dim3(1), dim3(1024), 0, st, src, (int)n, iou_thr, flags, rec, cands, ecap, keep_dev, result);
    else if (kind == OBB_KIND_POLY)
      hipLaunchKernelGGL(nms_small_kernel<1>, dim3(1), dim3(1024), 0, st, src, (int)n, iou_thr, flags, rec, cands, ecap, keep_dev, result);
    else if (f64)
      hipLaunchKernelGGL((nms_small_kernel<2, true>), dim3(1), dim3(1024), 0, st, src, (int)n, iou_thr, flags, rec, cands, ecap, keep_dev, result);
    else
      hipLaunchKernelGGL(nms_small_kernel<2>, dim3(1), dim3(1024), 0, st, src, (int)n, iou_thr, flags, rec, cands, ecap, keep_dev, result);
    OBB_LAUNCH_CHECK();
    return OBB_OK;
  }
  // threshold-aware pruning: the call's threshold, or per pair the smaller of the two boxes'
  const float prune_thr = thr_dev ? 0.0f : iou_thr;
  // ---- grid route (see nms_ginsert_kernel): float calls above the direct route's range, from the entry points that can
  // repeat a call (src.allow_grid), unless this thread's last try of the size class failed
  {
    const int64_t gknob = tuning(OBB_TUNE_NMS_GRID, -1);          // -1 auto, 0 never, 1 wherever it applies, 2 the same and give up at once (test hook)
    // (auto: up to kGridAutoMax boxes; profiles/r04_grid_vs_sorted.txt: 1.2 - 1.7 x the sorted route from 12 000 to 1.4 M
    // boxes, ungrouped and grouped)
    const bool grouped_call = src.mode != 0 || group_dev != nullptr;
    const bool grid = !direct && L.grid_log2 > 0 && n > direct_max && !f64 && src.allow_grid != 0 &&
                      gknob != 0 && (gknob >= 1 || n <= kGridAutoMax) && !grid_skip_take(n, kind + (grouped_call ? 4 : 0) + 8 * src.mode);
    q.grid = grid;
    if (grid) {
      const size_t gcells = (size_t)1 << L.grid_bits;
      const int gcoop = n < (1 << 18) ? 1 : 0;          // (the tail of the sorted route: compaction inside the resolve launch below 2^18 boxes, its own launch from there)
      // retained state (NmsRetain): only where the fused compaction is the call's tail; 2 = three launches
      const int retain = (src.retain != 0 && gcoop && tuning(OBB_TUNE_NMS_RETAIN, 1) != 0) ? src.retain : 0;
      q.retained = retain;
      NmsRetain *hdr = reinterpret_cast<NmsRetain *>(ws + L.retain);
      const unsigned long long seq = src.retain_seq;
      const unsigned int sig = retain_sig(L.grid_log2, L.grid_bits, grouped_call);
      NmsFrame *fr_cur = retain ? &hdr->frames[seq & 1ull] : frame, *fr_next = retain ? &hdr->frames[(seq + 1ull) & 1ull] : nullptr;
      uint32_t *gcnt = reinterpret_cast<uint32_t *>(ws + L.gcnt);
      NmsGridCtr *gctr = reinterpret_cast<NmsGridCtr *>(ws + L.gcnt - kGridCtrBytes);      // (the slot just below the cell counters)
      uint4 *gtab = reinterpret_cast<uint4 *>(ws + L.gtab);
      uint4 *gover = reinterpret_cast<uint4 *>(ws + L.gover);
      uint32_t *gbig = reinterpret_cast<uint32_t *>(ws + L.gbig);
      const int hbits = 7, grid_log2 = L.grid_log2;
      OBB_PHASE(0);
      if (retain != 2) {
        const size_t zbytes = L.hist - L.zero_begin;          // the call's counters ...
        uint4 *zero = reinterpret_cast<uint4 *>(ws + L.zero_begin);
        const size_t zero_vecs = zbytes / sizeof(uint4);
        uint4 *zero2 = reinterpret_cast<uint4 *>(gctr);        // ... the insert launch's three words and the cell counters behind them
        const size_t zero2_vecs = (kGridCtrBytes + gcells * sizeof(uint32_t)) / sizeof(uint4);
        unsigned zb = (unsigned)((zero_vecs + zero2_vecs + 1023) / 1024) + 1u;
        zb = zb > 256u ? 256u : zb;
        if (kind == OBB_KIND_RBOX) hipLaunchKernelGGL(nms_frame_kernel<0>, dim3(zb), dim3(256), 0, st, src, n, hbits, fr_cur, fr_next, zero, zero_vecs, zero2, zero2_vecs, grid_log2);
        else if (kind == OBB_KIND_POLY) hipLaunchKernelGGL(nms_frame_kernel<1>, dim3(zb), dim3(256), 0, st, src, n, hbits, fr_cur, fr_next, zero, zero_vecs, zero2, zero2_vecs, grid_log2);
        else hipLaunchKernelGGL(nms_frame_kernel<2>, dim3(zb), dim3(256), 0, st, src, n, hbits, fr_cur, fr_next, zero, zero_vecs, zero2, zero2_vecs, grid_log2);
        OBB_LAUNCH_CHECK();
      }
      const uint32_t gmask = (uint32_t)gcells - 1u;
      RetainArgs ra = {};
      if (retain == 2) {
        ra.hdr_magic = &hdr->magic; ra.expect = retain_magic(seq, sig);
        ra.frame_next = fr_next; ra.cnt = cnt; ra.ccnts = ccnts; ra.ucnts = ucnts; ra.blockcnt = blockcnt;
      }
      const unsigned iblocks = pblocks + (retain == 2 ? 1u : 0u);      // (+ the workgroup that takes the next call's frame)
      if (kind == OBB_KIND_RBOX) hipLaunchKernelGGL(nms_ginsert_kernel<0>, dim3(iblocks), dim3(256), 0, st, src, n, rec, ar, fr_cur, iou_thr, gcnt, gtab, ctr, gctr, gmask, gover, gbig, grid_big_cap(n), gknob == 2 ? 1 : 0, ra, hbits, grid_log2);
      else if (kind == OBB_KIND_POLY) hipLaunchKernelGGL(nms_ginsert_kernel<1>, dim3(iblocks), dim3(256), 0, st, src, n, rec, ar, fr_cur, iou_thr, gcnt, gtab, ctr, gctr, gmask, gover, gbig, grid_big_cap(n), gknob == 2 ? 1 : 0, ra, hbits, grid_log2);
      else hipLaunchKernelGGL(nms_ginsert_kernel<2>, dim3(iblocks), dim3(256), 0, st, src, n, rec, ar, fr_cur, iou_thr, gcnt, gtab, ctr, gctr, gmask, gover, gbig, grid_big_cap(n), gknob == 2 ? 1 : 0, ra, hbits, grid_log2);
      OBB_LAUNCH_CHECK();
      OBB_PHASE(1); OBB_PHASE(2);
      OBB_PHASE(3);
      // (waves beyond the chunks of the grid that exists leave on the frame record; a wave strides over the chunks when
      // the launch is capped)
      size_t gp_waves = gcells >> 1;                                    // the most chunks any number of cells <= gcells makes
      if (gcells >= kGridCpw4) { gp_waves = gcells >> 2; if (gp_waves < (kGridCpw4 >> 1)) gp_waves = kGridCpw4 >> 1; }
      if (gcells >= kGridCpw8) { gp_waves = gcells >> 3; if (gp_waves < (kGridCpw8 >> 2)) gp_waves = kGridCpw8 >> 2; }
      if (gp_waves > (size_t)kGridProbeWaves) gp_waves = kGridProbeWaves;
      const size_t gp_blocks = (gp_waves + kNmsWaves - 1) / kNmsWaves;
      const float *gthr = src.mode == 2 ? src.thr_table : nullptr;
      const int gthr_n = (int)(src.mult < (1 << 20) ? src.mult : (1 << 20));
      const unsigned long long *pmagic = retain == 2 ? &hdr->magic : nullptr;
      if (grouped_call)
        hipLaunchKernelGGL(nms_gprobe_kernel<true>, dim3((unsigned)gp_blocks), dim3(64 * kNmsWaves), 0, st,
                           rec, fr_cur, gcnt, gtab, (uint32_t)gcells, prune_thr, thr_dev ? 1 : 0, flags, edges, segcap, ccnts, cnt, gctr, gover, grid_over_cap(n), gthr, gthr_n, ar, gbig, grid_big_cap(n), (uint32_t)n, pmagic, ra.expect);
      else
        hipLaunchKernelGGL(nms_gprobe_kernel<false>, dim3((unsigned)gp_blocks), dim3(64 * kNmsWaves), 0, st,
                           rec, fr_cur, gcnt, gtab, (uint32_t)gcells, prune_thr, thr_dev ? 1 : 0, flags, edges, segcap, ccnts, cnt, gctr, gover, grid_over_cap(n), gthr, gthr_n, ar, gbig, grid_big_cap(n), (uint32_t)n, pmagic, ra.expect);
      OBB_LAUNCH_CHECK();
      OBB_PHASE(4);
      OBB_PHASE(5);
      RetainTail rt = {};
      if (retain) {
        rt.hdr_magic = &hdr->magic; rt.expect = retain == 2 ? ra.expect : 0ull; rt.next = retain_magic(seq + 1ull, sig);
        rt.gcnt = reinterpret_cast<uint4 *>(gcnt); rt.gcnt_vecs = gcells * sizeof(uint32_t) / sizeof(uint4);
      }
      hipLaunchKernelGGL(nms_resolve_kernel<false>, dim3(kShards * (gcoop ? 1 : resolve_parts())), dim3(kResolveThreads), 0, st, edges, segcap, cnt, ctr, ccnts, n, keep_dev, result, gcoop, blockcnt, ucnts, (int)tuning(OBB_TUNE_NMS_RESOLVE_TAIL, 1), gctr, rt);
      OBB_LAUNCH_CHECK();
      OBB_PHASE(6);
      if (!gcoop) {
        const unsigned cblocks = (unsigned)((n + 4095) / 4096);
        hipLaunchKernelGGL(nms_compact_kernel<4>, dim3(cblocks), dim3(1024), 0, st, cnt, n, blockcnt, keep_dev, ctr, ccnts, ucnts, result, gctr);
        OBB_LAUNCH_CHECK();
      }
      OBB_PHASE(7);
      return OBB_OK;
    }
  }
  // phase 0: prep (+ clears the counters) [+ quantise]
  OBB_PHASE(0);
  // cells ~ n/16 .. n/4, a power of 4 in [4096, 262144]; grouped calls of that size get the next
  // power: the group bucket takes bits away from space (a merged NMS of 1.25 M boxes in 240
  // groups had 6 bits of space left per bucket: ~80 boxes per cell, contended histogram words)
  int hbits = 6;
  while (hbits < 9 && (1ll << (2 * hbits)) * 16 < n) ++hbits;
  if (group_dev && n >= (1 << 18) && hbits < 9) ++hbits;
  // the direct path never touches the cell histogram; the sorted one clears the cells it uses
  const size_t zbytes = (direct ? L.scanflags : L.hist + ((size_t)4 << (2 * hbits))) - L.zero_begin;
  uint4 *zero = reinterpret_cast<uint4 *>(ws + L.zero_begin);
  const size_t zero_vecs = zbytes / sizeof(uint4);
  if (direct) {
    if (kind == OBB_KIND_RBOX)
      hipLaunchKernelGGL(nms_prep_kernel<0>, dim3(pblocks), dim3(256), 0, st, src, n, rec, iv, ar, iou_thr, zero, zero_vecs);
    else if (kind == OBB_KIND_POLY)
      hipLaunchKernelGGL(nms_prep_kernel<1>, dim3(pblocks), dim3(256), 0, st, src, n, rec, iv, ar, iou_thr, zero, zero_vecs);
    else if (f64)
      hipLaunchKernelGGL((nms_prep_kernel<2, true>), dim3(pblocks), dim3(256), 0, st, src, n, rec, iv, ar, iou_thr, zero, zero_vecs);
    else
      hipLaunchKernelGGL(nms_prep_kernel<2>, dim3(pblocks), dim3(256), 0, st, src, n, rec, iv, ar, iou_thr, zero, zero_vecs);
    OBB_LAUNCH_CHECK();
  } else {
    // frame (sampled, block 0) + clearing of the call's counters, histogram cells and tile-box words
    const int grid_log2 = kGridMaxLog2;                             // (the grid parameters are not used on this route)
    unsigned zb = (unsigned)((zero_vecs + 1023) / 1024) + 1u;     // (block 0 also takes the frame)
    zb = zb > 256u ? 256u : zb;
    if (kind == OBB_KIND_RBOX)
      hipLaunchKernelGGL(nms_frame_kernel<0>, dim3(zb), dim3(256), 0, st, src, n, hbits, frame, (NmsFrame *)nullptr, zero, zero_vecs, (uint4 *)nullptr, (size_t)0, grid_log2);
    else if (kind == OBB_KIND_POLY)
      hipLaunchKernelGGL(nms_frame_kernel<1>, dim3(zb), dim3(256), 0, st, src, n, hbits, frame, (NmsFrame *)nullptr, zero, zero_vecs, (uint4 *)nullptr, (size_t)0, grid_log2);
    else if (f64)
      hipLaunchKernelGGL((nms_frame_kernel<2, true>), dim3(zb), dim3(256), 0, st, src, n, hbits, frame, (NmsFrame *)nullptr, zero, zero_vecs, (uint4 *)nullptr, (size_t)0, grid_log2);
    else
      hipLaunchKernelGGL(nms_frame_kernel<2>, dim3(zb), dim3(256), 0, st, src, n, hbits, frame, (NmsFrame *)nullptr, zero, zero_vecs, (uint4 *)nullptr, (size_t)0, grid_log2);
    OBB_LAUNCH_CHECK();
    if (kind == OBB_KIND_RBOX)
      hipLaunchKernelGGL(nms_prepq_kernel<0>, dim3(pblocks), dim3(256), 0, st, src, n, rec, qb, ar, ckey, crank, hist, hbits, frame, iou_thr);
    else if (kind == OBB_KIND_POLY)
      hipLaunchKernelGGL(nms_prepq_kernel<1>, dim3(pblocks), dim3(256), 0, st, src, n, rec, qb, ar, ckey, crank, hist, hbits, frame, iou_thr);
    else if (f64)
      hipLaunchKernelGGL((nms_prepq_kernel<2, true>), dim3(pblocks), dim3(256), 0, st, src, n, rec, qb, ar, ckey, crank, hist, hbits, frame, iou_thr);
    else
      hipLaunchKernelGGL(nms_prepq_kernel<2>, dim3(pblocks), dim3(256), 0, st, src, n, rec, qb, ar, ckey, crank, hist, hbits, frame, iou_thr);
    OBB_LAUNCH_CHECK();
  }
  if (direct) {
    OBB_PHASE(1); OBB_PHASE(2);
    OBB_PHASE(3);
    // 32 rows per wave; fewer (more waves) while that leaves most SIMDs without one
    int rpw = kSubRows;
    while (rpw > 4 && ((n + rpw - 1) / rpw) * (long long)(nt + 1) / 2 < 2048) rpw >>= 1;
    const unsigned nparts = (unsigned)((n + rpw - 1) / rpw);
    hipLaunchKernelGGL(nms_scan_direct_kernel, dim3((nparts + kNmsWaves - 1) / kNmsWaves, (unsigned)nt), dim3(64 * kNmsWaves), 0, st,
                       iv, ar, rec, (uint32_t)n, rpw, prune_thr, thr_dev ? 1 : 0, flags, edges, segcap, ccnts, cnt, klo);
    OBB_LAUNCH_CHECK();
    OBB_PHASE(4);
    OBB_PHASE(5);
    hipLaunchKernelGGL(nms_resolve_kernel<true>, dim3(kShards), dim3(kResolveThreadsCompact), 0, st, edges, segcap, cnt, ctr, ccnts, n, keep_dev, result, 0, blockcnt, ucnts, (int)tuning(OBB_TUNE_NMS_RESOLVE_TAIL, 1), (NmsGridCtr *)nullptr, RetainTail{});
    OBB_LAUNCH_CHECK();
    OBB_PHASE(6);
    OBB_PHASE(7);
  } else {
    // phase 1: spatial sort = cell scan (inside every scatter block up to 16384 cells) + scatter of the packed
    // intervals and per-position records to their sorted positions
    OBB_PHASE(1);
    {
      unsigned sblocks = (unsigned)((n + 1023) / 1024);
      if (sblocks > 2048u) sblocks = 2048u;
      const int64_t npos = (int64_t)nt * kTile;
      if (hbits == 6) {
        hipLaunchKernelGGL(nms_sortscatter_kernel<4096>, dim3(sblocks), dim3(1024), 0, st, hist, qb, ar, ckey, crank, n, npos, group_dev, thr_dev, qs, pf, pg, pt);
      } else if (hbits == 7) {
        hipLaunchKernelGGL(nms_sortscatter_kernel<16384>, dim3(sblocks), dim3(1024), 0, st, hist, qb, ar, ckey, crank, n, npos, group_dev, thr_dev, qs, pf, pg, pt);
      } else {
        if (hbits == 8) hipLaunchKernelGGL(nms_cellscan_kernel<64>, dim3(1), dim3(1024), 0, st, hist);
        else hipLaunchKernelGGL(nms_cellscan_big_kernel, dim3((1u << (2 * hbits)) / kScanBlockCells), dim3(256), 0, st, hist, scanflags);
        OBB_LAUNCH_CHECK();
        hipLaunchKernelGGL(nms_sortscatter_kernel<0>, dim3(sblocks), dim3(1024), 0, st, hist, qb, ar, ckey, crank, n, npos, group_dev, thr_dev, qs, pf, pg, pt);
      }
      OBB_LAUNCH_CHECK();
    }
    // phase 2: tile boxes + unit culling (one launch below kSuperMinTiles tiles)
    OBB_PHASE(2);
    if (nt < kSuperMinTiles) {
      hipLaunchKernelGGL(nms_cullbox_kernel, dim3((unsigned)nt), dim3(256), 0, st, qs, n, pg, frame, nt, tbox, tgr, units, ucap / kShards, ucnts, ticket);
      OBB_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(nms_tilebox_kernel, dim3((unsigned)nt), dim3(kTile), 0, st, qs, n, tbox, sbox, pg, frame, tgr, sgr, super);
      OBB_LAUNCH_CHECK();
      hipLaunchKernelGGL(nms_cull_kernel, dim3((unsigned)nt, (unsigned)((nt + kCullChunk - 1) / kCullChunk)), dim3(256), 0, st, tbox, sbox, group_dev ? tgr : nullptr, sgr, nt, units, ucap / kShards, ucnts, super);
      OBB_LAUNCH_CHECK();
    }
    // phase 3: scan (broad phase)
    OBB_PHASE(3);
    {
      const long long maxblocks = ((long long)L.unit_worst + kNmsWaves - 1) / kNmsWaves;
      // tools/exp_scan_blocks.py: the in-line kernel (4 waves/SIMD) is best at one resident grid; the
      // listing one (8 waves/SIMD) gains 3-8 % from two to four grids' worth of smaller shares
      const long long pb = tuning(OBB_TUNE_NMS_SCAN_BLOCKS, inline_clip ? kScanBlocks : (n > 262144 ? 4 * kScanBlocks : 2 * kScanBlocks));
      const unsigned gb = (unsigned)(maxblocks < pb ? maxblocks : pb);
      if (inline_clip)
        hipLaunchKernelGGL(nms_scan_kernel<true>, dim3(gb), dim3(64 * kNmsWaves), 0, st, qs, pf, pg, pt, cellpx, prune_thr, tbox, nt, units, ucap / kShards, ucnts, cands, segcap, ccnts, rec, flags, edges, cnt, klo);
      else
        hipLaunchKernelGGL(nms_scan_kernel<false>, dim3(gb), dim3(64 * kNmsWaves), 0, st, qs, pf, pg, pt, cellpx, prune_thr, tbox, nt, units, ucap / kShards, ucnts, cands, segcap, ccnts, rec, flags, edges, cnt, klo);
      OBB_LAUNCH_CHECK();
    }
    // phase 4: clip (narrow phase) -- its own launch only where the scan lists its candidates
    OBB_PHASE(4);
    if (!inline_clip) {
      hipLaunchKernelGGL(nms_clip_kernel, dim3(kClipBlocksPerShard, kShards), dim3(256), 0, st, cands, segcap, rec, flags, edges, ctr, ccnts, cnt, klo);
      OBB_LAUNCH_CHECK();
    }
    // phase 5: resolve
    OBB_PHASE(5);
    // (below 2^18 boxes the compaction is the tail of the resolve launch: its 64 workgroups take n / 64 words each)
    const int coop = n < (1 << 18) ? (int)tuning(OBB_TUNE_NMS_FUSED_COMPACT, 1) : 0;   // (2: waiters give up at once -- the fallback under test)
    hipLaunchKernelGGL(nms_resolve_kernel<false>, dim3(kShards * (coop ? 1 : resolve_parts())), dim3(kResolveThreads), 0, st, edges, segcap, cnt, ctr, ccnts, n, keep_dev, result, coop, blockcnt, ucnts, (int)tuning(OBB_TUNE_NMS_RESOLVE_TAIL, 1), (NmsGridCtr *)nullptr, RetainTail{});
    OBB_LAUNCH_CHECK();
    // phase 6: compaction (one pass: every block publishes its count and sums its predecessors')
    OBB_PHASE(6);
    if (coop) {
    } else if (n >= (1 << 18)) {
      const unsigned cblocks = (unsigned)((n + 4095) / 4096);
      hipLaunchKernelGGL(nms_compact_kernel<4>, dim3(cblocks), dim3(1024), 0, st, cnt, n, blockcnt, keep_dev, ctr, ccnts, ucnts, result, (const NmsGridCtr *)nullptr);
    } else {
      const unsigned cblocks = (unsigned)((n + 1023) / 1024);
      hipLaunchKernelGGL(nms_compact_kernel<1>, dim3(cblocks), dim3(1024), 0, st, cnt, n, blockcnt, keep_dev, ctr, ccnts, ucnts, result, (const NmsGridCtr *)nullptr);
    }
    OBB_LAUNCH_CHECK();
    OBB_PHASE(7);
  }
#undef OBB_PHASE
  return OBB_OK;
}

// the plain entry points: rows of S floats, group / threshold arrays
static int nms_enqueue(const char *who, const float *dets_dev, int64_t n, int kind, float iou_thr,
                       const int32_t *group_dev, const float *thr_dev, int flags_in,
                       int64_t *keep_dev, NmsHostInfo *result, void *workspace_dev,
                       size_t workspace_bytes, hipStream_t st, NmsEnqueued &q, bool allow_grid = false,
                       int retain = 0, unsigned long long retain_seq = 0ull) {
  NmsSource src = {};
  src.dets = dets_dev;
  src.pitch = kind == OBB_KIND_RBOX ? 6 : (kind == OBB_KIND_POLY ? 9 : 5);
  src.group = group_dev;
  src.thrs = thr_dev;
  src.allow_grid = allow_grid ? 1 : 0;
  src.retain = retain; src.retain_seq = retain_seq;
  return nms_enqueue_source(who, src, n, kind, iou_thr, flags_in, keep_dev, result, workspace_dev, workspace_bytes, st, q);
}

// Waiting for a synchronous call.  Its last kernel's last act is the 64-byte result record, written straight into pinned
// host memory: the host fills the record with ones before it queues the call and then WATCHES it -- every word different
// from the fill = the record is here (no field of a record is all ones; the words may arrive in any order) -- instead of
// sleeping in hipStreamSynchronize, whose wake-up costs several microseconds of a 60 us call.  The stream may still be
// retiring that kernel when the entry point returns: whatever is queued behind it is ordered as usual.  A record that has
// not come after ~1 ms (a long call, a fault) is waited for the old way, which also surfaces the stream's error.
static void nms_record_arm(NmsHostInfo *host) {
  volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(host);
  for (size_t i = 0; i < sizeof(NmsHostInfo) / 8; ++i) w[i] = ~0ull;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
static int nms_record_wait(NmsHostInfo *host, hipStream_t st, bool must_sync) {
  if (!must_sync) {
    volatile unsigned long long *w = reinterpret_cast<volatile unsigned long long *>(host);
    for (int spin = 0; spin < 200000; ++spin) {
      bool all = true;
      for (size_t i = 0; i < sizeof(NmsHostInfo) / 8; ++i) all = all && (w[i] != ~0ull);
      if (all) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return OBB_OK; }
      __builtin_ia32_pause();
    }
  }
  OBB_HIP_TRY(hipStreamSynchronize(st));
  return OBB_OK;
}
static_assert(sizeof(NmsHostInfo) % 8 == 0 && sizeof(NmsHostInfo) <= 64, "the record is watched word by word");

// ---- retained workspaces (OBB_NMS_RETAIN; NmsRetain): the host's note of what a workspace's header says.  Per calling
// thread, by workspace pointer; the device checks the header too, so a note that is wrong (another thread used the
// workspace, the memory was handed out again) costs a repeated call, not a result.
struct RetainSlot { const void *ws; unsigned int sig; unsigned long long seq; bool valid; unsigned long long used; };
constexpr int kRetainSlots = 8;
static thread_local RetainSlot g_retain[kRetainSlots] = {};
static thread_local unsigned long long g_retain_clock = 0ull;
static thread_local int g_last_retained = 0;
static RetainSlot *retain_find(const void *ws) {
  for (RetainSlot &r : g_retain) if (r.ws == ws) return &r;
  return nullptr;
}
static RetainSlot *retain_slot(const void *ws) {
  RetainSlot *r = retain_find(ws);
  if (!r) {
    r = &g_retain[0];
    for (RetainSlot &c : g_retain) if (c.used < r->used) r = &c;     // least recently used (an empty slot: 0)
    // (a sequence number nobody has used on this memory: the clock in the upper bits -- a stale header of an earlier
    //  owner of the address does not match)
    *r = RetainSlot{ws, 0u, (++g_retain_clock) << 32, false, 0ull};
  }
  r->used = ++g_retain_clock;
  return r;
}
// any call that writes a workspace outside the retained protocol ends what the thread believes about it
void nms_retain_forget(const void *ws) { if (RetainSlot *r = retain_find(ws)) r->valid = false; }

// one pinned, device-mapped result record per calling thread (synchronous entry points)
static int nms_pinned_record(NmsHostInfo *&host, NmsHostInfo *&dev) {
  static thread_local NmsHostInfo *hinfo = nullptr;
  if (!hinfo) {
    void *p = nullptr;
    OBB_HIP_TRY(hipHostMalloc(&p, 256, hipHostMallocPortable | hipHostMallocMapped));
    hinfo = static_cast<NmsHostInfo *>(p);
  }
  host = hinfo;
  OBB_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), hinfo, 0));
  return OBB_OK;
}

}  // namespace obb

extern "C" int obb_nms_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                           int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                           size_t workspace_bytes, void *stream) {
  return obb_nms_ex_f32(dets_dev, n, kind, iou_thr, nullptr, nullptr, 0, keep_dev, info_host,
                        workspace_dev, workspace_bytes, stream);
}

extern "C" int obb_nms_grouped_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                                   const int32_t *group_dev, const float *thr_dev,
                                   int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                                   size_t workspace_bytes, void *stream) {
  return obb_nms_ex_f32(dets_dev, n, kind, iou_thr, group_dev, thr_dev, 0, keep_dev, info_host,
                        workspace_dev, workspace_bytes, stream);
}

extern "C" int obb_nms_ex_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                              const int32_t *group_dev, const float *thr_dev, int flags,
                              int64_t *keep_dev, int64_t *info_host, void *workspace_dev,
                              size_t workspace_bytes, void *stream) {
  using namespace obb;
  if (info_host) { info_host[0] = 0; info_host[1] = 0; info_host[2] = 0; info_host[3] = 0; }
  int rc = nms_validate("obb_nms_f32", dets_dev, n, kind, iou_thr, thr_dev != nullptr, flags, keep_dev, workspace_dev);
  if (rc != OBB_OK) return rc;
  if (n == 0) return OBB_OK;
  NmsHostInfo *hinfo = nullptr, *hinfo_dev = nullptr;
  rc = nms_pinned_record(hinfo, hinfo_dev);
  if (rc != OBB_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  NmsEnqueued q;
  const bool watch = tuning(OBB_TUNE_NMS_WATCH_RECORD, 1) != 0;
  // retained workspace (OBB_NMS_RETAIN): 2 = the three-launch call on the state the previous call left, 1 = seed it
  const bool want_retain = (flags & OBB_NMS_RETAIN) != 0 && tuning(OBB_TUNE_NMS_RETAIN, 1) != 0;
  RetainSlot *slot = want_retain ? retain_slot(workspace_dev) : nullptr;
  unsigned int sig = 0u;
  if (slot) { int g = 0, bits = 0; grid_geometry(n, g, bits); sig = retain_sig(g, bits, group_dev != nullptr); }
  if (!slot) nms_retain_forget(workspace_dev);
  int mode = slot ? ((slot->valid && slot->sig == sig) ? 2 : 1) : 0;
  g_last_retained = 0;
  // (every launch sequence that ran to its end on a retained workspace -- whatever the verdict -- has left the header
  //  naming the next call: the note follows)
  auto retain_done = [&]() {
    if (!slot) return;
    if (q.retained && (hinfo->route & 0xFFu) != kRouteUnseeded) { slot->valid = true; slot->sig = sig; slot->seq += 1ull; g_last_retained = q.retained; }
    else slot->valid = false;      // another route wrote the workspace (or nothing ran): seed again next time
  };
  nms_record_arm(hinfo);
  rc = nms_enqueue("obb_nms_f32", dets_dev, n, kind, iou_thr, group_dev, thr_dev, flags, keep_dev, hinfo_dev,
                   workspace_dev, workspace_bytes, st, q, true, mode, slot ? slot->seq : 0ull);
  if (rc != OBB_OK) { if (slot) slot->valid = false; return rc; }
  rc = nms_record_wait(hinfo, st, (q.timed && g_prof.level >= 2) || !watch);
  if (rc != OBB_OK) { if (slot) slot->valid = false; return rc; }
  bool reseed = q.retained == 2 && (hinfo->route & 0xFFu) == kRouteUnseeded;    // somebody else wrote the workspace: nothing was computed
  retain_done();
  // a retained call on a crowded grid: the frame its predecessor left may be the reason -- once more with a frame of its own
  if (!reseed && q.retained == 2 && q.grid && (hinfo->route & 0xFFu) == 3u) reseed = true;
  if (reseed) {
    q = NmsEnqueued();
    OBB_HIP_TRY(hipStreamSynchronize(st));
    nms_record_arm(hinfo);
    rc = nms_enqueue("obb_nms_f32", dets_dev, n, kind, iou_thr, group_dev, thr_dev, flags, keep_dev, hinfo_dev,
                     workspace_dev, workspace_bytes, st, q, true, 1, slot->seq);
    if (rc != OBB_OK) { slot->valid = false; return rc; }
    rc = nms_record_wait(hinfo, st, (q.timed && g_prof.level >= 2) || !watch);
    if (rc != OBB_OK) { slot->valid = false; return rc; }
    retain_done();
  }
  g_last_route = q.grid ? 4 : (q.fused ? 0 : (q.direct ? 1 : 2));
  if (q.grid && (hinfo->route & 0xFFu) == 3u) {
    g_last_route = 5;
    // the grid route gave up (a crowded cell, a box over too many cells): the same call on the sorted route, and this
    // thread's next calls of the size class go there directly
    grid_skip_put(n, kind + ((group_dev != nullptr) ? 4 : 0));
    q = NmsEnqueued();
    OBB_HIP_TRY(hipStreamSynchronize(st));       // (the call that gave up has retired before the record is armed again)
    nms_record_arm(hinfo);
    rc = nms_enqueue("obb_nms_f32", dets_dev, n, kind, iou_thr, group_dev, thr_dev, flags, keep_dev, hinfo_dev,
                     workspace_dev, workspace_bytes, st, q, false);
    if (rc != OBB_OK) return rc;
    rc = nms_record_wait(hinfo, st, (q.timed && g_prof.level >= 2) || !watch);
    if (rc != OBB_OK) return rc;
    g_last_retained = 0;
  }
  const NmsHostInfo h = *hinfo;
  NmsProfile &prof = g_prof;
  if (q.timed) {
    // (level 1 brackets the dominant kernel only: both events lie in front of the call's last launch, so the call watched its
    //  record like an untimed one -- the later event has been reached; waiting on IT is all the synchronisation there is)
    if (prof.level < 2 && !q.fused) OBB_HIP_TRY(hipEventSynchronize(prof.ev[4]));
    for (int k = 0; k < kPhases; ++k) {
      prof.ms[k] = 0.0f;
      if (!q.fused && (prof.level >= 2 || k == 3)) OBB_HIP_TRY(hipEventElapsedTime(&prof.ms[k], prof.ev[k], prof.ev[k + 1]));
    }
    prof.have = true;
  }
  static_assert(kShards == 64, "the counter folds use one wave");
  int64_t info_local[4];
  int64_t *info = info_host ? info_host : info_local;
  rc = nms_verdict(h, q, info);
  cap_hint_note(n, kind, rc, rc == OBB_EWORKSPACE ? (unsigned long long)info[3] : 0ull, h.cand_total > h.edges ? h.cand_total : h.edges);
  return rc;
}

extern "C" int obb_nms_hbb_f64(const double *dets_dev, int64_t n, float iou_thr, const int32_t *group_dev,
                               const float *thr_dev, int flags, int64_t *keep_dev, int64_t *info_host,
                               void *workspace_dev, size_t workspace_bytes, void *stream) {
  using namespace obb;
  if (info_host) { info_host[0] = 0; info_host[1] = 0; info_host[2] = 0; info_host[3] = 0; }
  int rc = nms_validate("obb_nms_hbb_f64", reinterpret_cast<const float *>(dets_dev), n, OBB_KIND_HBB, iou_thr, thr_dev != nullptr, flags,
                        keep_dev, workspace_dev);
  if (rc != OBB_OK) return rc;
  if (flags & OBB_NMS_RETAIN) return set_err(OBB_EINVAL, "obb_nms_hbb_f64: OBB_NMS_RETAIN applies to obb_nms_ex_f32");
  if (n == 0) return OBB_OK;
  NmsHostInfo *hinfo = nullptr, *hinfo_dev = nullptr;
  rc = nms_pinned_record(hinfo, hinfo_dev);
  if (rc != OBB_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  NmsSource src = {};
  src.dets64 = dets_dev; src.pitch = 5; src.group = group_dev; src.thrs = thr_dev;
  nms_retain_forget(workspace_dev);
  NmsEnqueued q;
  rc = nms_enqueue_source("obb_nms_hbb_f64", src, n, OBB_KIND_HBB, iou_thr, flags, keep_dev, hinfo_dev, workspace_dev, workspace_bytes, st, q);
  if (rc != OBB_OK) return rc;
  OBB_HIP_TRY(hipStreamSynchronize(st));
  const NmsHostInfo h = *hinfo;
  int64_t info_local[4];
  int64_t *info = info_host ? info_host : info_local;
  rc = nms_verdict(h, q, info);
  cap_hint_note(n, OBB_KIND_HBB, rc, rc == OBB_EWORKSPACE ? (unsigned long long)info[3] : 0ull, h.cand_total > h.edges ? h.cand_total : h.edges);
  return rc;
}

extern "C" int obb_nms_async_f32(const float *dets_dev, int64_t n, int kind, float iou_thr,
                                 const int32_t *group_dev, const float *thr_dev, int flags,
                                 int64_t *keep_dev, void *result_dev, void *workspace_dev,
                                 size_t workspace_bytes, void *stream) {
  using namespace obb;
  if (!result_dev) return set_err(OBB_EINVAL, "obb_nms_async_f32: null result record");
  hipStream_t st = (hipStream_t)stream;
  int rc = nms_validate("obb_nms_async_f32", dets_dev, n, kind, iou_thr, thr_dev != nullptr, flags, keep_dev, workspace_dev);
  if (rc != OBB_OK) return rc;
  if (n == 0) {   // an all-zero record: K = 0, route "fused" (its check passes for 0 edges)
    OBB_HIP_TRY(hipMemsetAsync(result_dev, 0, OBB_NMS_RESULT_BYTES, st));
    return OBB_OK;
  }
  if (flags & OBB_NMS_RETAIN) return set_err(OBB_EINVAL, "obb_nms_async_f32: OBB_NMS_RETAIN needs the synchronous entry point (the host reads the record)");
  nms_retain_forget(workspace_dev);
  NmsEnqueued q;
  return nms_enqueue("obb_nms_async_f32", dets_dev, n, kind, iou_thr, group_dev, thr_dev, flags, keep_dev,
                     static_cast<NmsHostInfo *>(result_dev), workspace_dev, workspace_bytes, st, q);
}

extern "C" int obb_nms_async_result(const void *result_host, int64_t n, size_t workspace_bytes,
                                    int64_t *info_host) {
  using namespace obb;
  if (!result_host || !info_host) return set_err(OBB_EINVAL, "obb_nms_async_result: null pointer");
  info_host[0] = info_host[1] = info_host[2] = info_host[3] = 0;
  if (n == 0) return OBB_OK;
  NmsHostInfo h;
  memcpy(&h, result_host, sizeof(h));
  if ((h.route & 0xFFu) == 3u) {
    // the grid route gave up on this scene: the caller repeats the call (or the tiled phase) as it is, and this thread's
    // next calls of the size class take the sorted route from the start
    grid_skip_put(n, (int)(h.route >> 8) - 1);
    return set_err(OBB_ERETRY, "the grid route gave up on this scene (a crowded cell): repeat the call as it is");
  }
  NmsLayout L;
  NmsEnqueued q;
  int rc = nms_capacities(n, workspace_bytes, L, q);
  if (rc != OBB_OK) return rc;
  rc = nms_verdict(h, q, info_host);
  // (the record does not say which kind it was: rotated boxes -- what the async form is used for in the tiled phases and
  // the detector's queued calls; a hint under the wrong kind only costs one repeated call)
  cap_hint_note(n, OBB_KIND_RBOX, rc, rc == OBB_EWORKSPACE ? (unsigned long long)info_host[3] : 0ull,
                h.cand_total > h.edges ? h.cand_total : h.edges);
  return rc;
}

extern "C" int obb_nms_last_route(void) { return obb::g_last_route; }
extern "C" int obb_nms_last_retained(void) { return obb::g_last_retained; }
extern "C" int obb_nms_retain_forget(const void *workspace_dev) {
  if (workspace_dev) obb::nms_retain_forget(workspace_dev);
  else for (obb::RetainSlot &r : obb::g_retain) r.valid = false;
  return OBB_OK;
}

extern "C" int obb_nms_profile_enable(int on) {
  obb::g_prof.on = on != 0;
  obb::g_prof.level = on;
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

#ifdef OBB_RESOLVE_STAMPS
extern "C" __attribute__((visibility("default"))) int obb_debug_resolve_stamps(unsigned long long *host, int words) {
  if (words > 64 * 64 * 2) words = 64 * 64 * 2;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_resolve_stamps), (size_t)words * 8) == hipSuccess ? 0 : -1;
}
#endif
