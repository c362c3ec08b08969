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
//   obb_rows_route_f32           rows -> stable buckets by (image id % world) + per-bucket counts  (routed layout)
//   obb_rows_restore_f32         the sorted runs of several ranks -> one run in ascending key      (all layouts)
// Pure data movement plus one add and one divide per coordinate, evaluated exactly like the
// tensor expressions they replace (aidet_amd/dota_merge.py): results are bit-identical.
#include "obb_common.h"
#include "obb_nms_internal.h"

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
                                                                   float *__restrict__ out,
                                                                   int64_t *__restrict__ patch_of_row) {
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= k) return;
  const int64_t src = keep ? keep[i] : i;
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = make_float4(0.0f, 0.0f, -1.0f, -1.0f);
  if (patch_of_row) patch_of_row[i] = (src >= 0 && src < n) ? patch_index[src] : -1;
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
__global__ __launch_bounds__(kMergeThreads) void merge_split_kernel(const float *__restrict__ rows, int cols, int ld,
                                                                    int64_t n, int64_t num_classes,
                                                                    const float *__restrict__ class_thrs,
                                                                    float default_thr,
                                                                    float *__restrict__ boxes,
                                                                    int32_t *__restrict__ groups,
                                                                    float *__restrict__ thrs) {
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= n) return;
  const int nbox = cols - 2;
  const float *r = rows + i * ld;
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


// ---- routing and order restoration (config 5 over several ranks) -------------------------------------------------
// The tensor-op forms (a stable 16-bit argsort + gather + bincount, aidet_amd/dota_merge.py) are ~25 small
// launches, 0.3 ms of a 0.67 ms pass in the 8-rank simulation; both are counting sorts with structure:
//   route:   <= 64 buckets (destination rank) -> per-block bucket counts, one scan, a stable scatter;
//   restore: every rank's rows arrive as ONE run in non-decreasing key, so the rows of a (key, source) pair are
//            contiguous: position = start[(key, source)] + distance to the first row of the pair.
constexpr int kRouteRows = 1024;        // rows per block (4 per thread)
constexpr int kRouteMaxWorld = 64;
constexpr int kScanThreads = 1024;

struct SegStarts { int64_t s[kRouteMaxWorld + 1]; };   // row range of every source in the gathered / received rows

__device__ __forceinline__ int route_bucket(float id, int world) {
  const int64_t v = (int64_t)id;                     // (truncation, like Tensor.to(int64))
  const int64_t m = v % (int64_t)world;
  return (int)(m < 0 ? m + world : m);               // (the sign of the divisor, like Tensor.__mod__)
}

__global__ __launch_bounds__(256) void route_count_kernel(const float *__restrict__ rows, int cols, int id_col,
                                                          int64_t n, int world, const int64_t *__restrict__ key,
                                                          int nblocks, int *__restrict__ blockcnt,
                                                          int *__restrict__ blockviol) {
  __shared__ int hist[kRouteMaxWorld];
  __shared__ int viol;
  const int tid = threadIdx.x;
  if (tid < kRouteMaxWorld) hist[tid] = 0;
  if (tid == 0) viol = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRouteRows;
  int bad = 0;
#pragma unroll
  for (int u = 0; u < kRouteRows / 256; ++u) {
    const int64_t i = base + u * 256 + tid;
    if (i < n) {
      atomicAdd(&hist[route_bucket(rows[i * cols + id_col], world)], 1);
      if (key && i > 0 && key[i] < key[i - 1]) ++bad;
    }
  }
  if (bad) atomicAdd(&viol, bad);
  __syncthreads();
  if (tid < world) blockcnt[(int64_t)tid * nblocks + blockIdx.x] = hist[tid];
  if (tid == 0) blockviol[blockIdx.x] = viol;
}

// exclusive scan of a[0, m) in place by ONE workgroup (m is a few 10^4: bucket x block counts, key x source counts)
constexpr int kScanLds = 16384;         // counts that are scanned in LDS (64 KiB): coalesced loads and stores
__device__ __forceinline__ int block_scan_inplace(int *__restrict__ a, int64_t m, int *part /* LDS, kScanThreads / 64 + 1 */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tiles of kScanLds counts with a running carry.  (Every thread walking a chunk of the WHOLE array in global memory twice --
  // dependent round trips at a stride of the chunk -- took 11 us for the 12 808 (patch, source) counts of an 8-rank pass and
  // 157 us for the 102 400 of the 128-image scene: 17 % of what one rank of eight computes there.)
  // (thread t owns the kPer consecutive counts [t kPer, (t + 1) kPer) of a tile: four 16-byte loads, the sums and the prefix in
  // registers, four 16-byte stores -- the version that staged the tile in LDS and walked it there at a stride of kPer words
  // took 6 us for 12 808 counts and 15 us for a full tile)
  constexpr int kPer = kScanLds / kScanThreads;
  static_assert(kPer % 4 == 0, "16-byte loads");
  int carry = 0;                                           // (block-uniform)
  for (int64_t base = 0; base < m; base += kScanLds) {
    const int mm = (int)(m - base < (int64_t)kScanLds ? m - base : (int64_t)kScanLds);
    int *t = a + base;
    const int lo = tid * kPer;
    const bool wide = ((reinterpret_cast<uintptr_t>(t) & 15u) == 0u) && lo + kPer <= mm;      // (a whole, aligned run of this thread)
    int v[kPer];
    if (wide) {
#pragma unroll
      for (int u = 0; u < kPer / 4; ++u) {
        const int4 q = reinterpret_cast<const int4 *>(t + lo)[u];
        v[4 * u] = q.x; v[4 * u + 1] = q.y; v[4 * u + 2] = q.z; v[4 * u + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int u = 0; u < kPer; ++u) v[u] = lo + u < mm ? t[lo + u] : 0;
    }
    int sum = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) sum += v[u];
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    int pre = carry + inc - sum, total = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) { const int x = part[w]; pre += (w < wave) ? x : 0; total += x; }
#pragma unroll
    for (int u = 0; u < kPer; ++u) { const int x = v[u]; v[u] = pre; pre += x; }
    if (wide) {
#pragma unroll
      for (int u = 0; u < kPer / 4; ++u) reinterpret_cast<int4 *>(t + lo)[u] = make_int4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
    } else {
#pragma unroll
      for (int u = 0; u < kPer; ++u) if (lo + u < mm) t[lo + u] = v[u];
    }
    carry += total;
    __syncthreads();                                       // (part is reused by the next tile)
  }
  return carry;
}

__global__ __launch_bounds__(kScanThreads) void route_scan_kernel(int *__restrict__ blockcnt, int world, int nblocks,
                                                                  const int *__restrict__ blockviol,
                                                                  int64_t *__restrict__ counts) {
  __shared__ int part[kScanThreads / 64 + 1];
  __shared__ int vsum;
  const int tid = threadIdx.x;
  if (tid == 0) vsum = 0;
  __syncthreads();
  int v = 0;
  for (int j = tid; j < nblocks; j += kScanThreads) v += blockviol[j];
  if (v) atomicAdd(&vsum, v);
  const int total = block_scan_inplace(blockcnt, (int64_t)world * nblocks, part);
  if (tid < world) {
    const int lo = blockcnt[(int64_t)tid * nblocks];
    const int hi = tid + 1 < world ? blockcnt[(int64_t)(tid + 1) * nblocks] : total;
    counts[tid] = hi - lo;
  }
  if (tid == 0) counts[world] = vsum;
}

__global__ __launch_bounds__(256) void route_scatter_kernel(const float *__restrict__ rows, int cols, int id_col,
                                                            int64_t n, int world, const int64_t *__restrict__ key,
                                                            int nblocks, const int *__restrict__ offs,
                                                            float *__restrict__ out) {
  constexpr int kU = kRouteRows / 256;
  __shared__ int wc[kU * 4 + 1][kRouteMaxWorld];     // rows of bucket b in (pass u, wave w); after the prefix: before it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < (kU * 4 + 1) * kRouteMaxWorld; j += 256) (&wc[0][0])[j] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRouteRows;
  int bk[kU], rk[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = base + u * 256 + tid;
    const bool valid = i < n;
    bk[u] = valid ? route_bucket(rows[i * cols + id_col], world) : -1;
    rk[u] = 0;
    uint64_t rem = __ballot(valid);
    while (rem) {                                    // one round per bucket present in the wave
      const int lead = __builtin_ctzll(rem);
      const int bl = __builtin_amdgcn_readlane(bk[u], lead);
      const uint64_t m = __ballot(bk[u] == bl);
      if (bk[u] == bl) rk[u] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == lead) wc[u * 4 + wave][bl] = __popcll(m);
      rem &= ~m;
    }
  }
  __syncthreads();
  if (tid < world) {                                 // exclusive prefix over (pass, wave) in row order, per bucket
    int run = offs[(int64_t)tid * nblocks + blockIdx.x];
    for (int j = 0; j < kU * 4; ++j) { const int v = wc[j][tid]; wc[j][tid] = run; run += v; }
  }
  __syncthreads();
  const int oc = cols + 1;
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = base + u * 256 + tid;
    if (i >= n) continue;
    const int64_t pos = (int64_t)wc[u * 4 + wave][bk[u]] + rk[u];
    const float *src = rows + i * cols;
    float *dst = out + pos * oc;
    for (int c = 0; c < cols; ++c) dst[c] = src[c];
    dst[cols] = key ? (float)key[i] : 0.0f;
  }
}

__device__ __forceinline__ int seg_of(const SegStarts &sg, int world, int64_t i) {
  int lo = 0, hi = world - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (sg.s[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// cnt[key * world + source] += 1 per real row (one atomic per run and wave), first[...] = the run's first row;
// cnt[m] counts the rows that break the promise (a key smaller than its predecessor's in the same source)
__global__ __launch_bounds__(256) void restore_mark_kernel(const float *__restrict__ rows, int cols, int key_col,
                                                           int64_t n, SegStarts sg, int world, int key_max,
                                                           int64_t m, int *__restrict__ cnt,
                                                           int64_t *__restrict__ first) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t idx = -1;
  bool boundary = false;
  if (i < n) {
    const int s = seg_of(sg, world, i);
    const float kf = rows[i * cols + key_col];
    if (kf >= 0.0f && kf <= (float)key_max) {
      const int k = (int)kf;
      idx = (int64_t)k * world + s;
      boundary = true;
      if (i > sg.s[s]) {
        const float pf = rows[(i - 1) * cols + key_col];
        if (pf >= 0.0f && pf <= (float)key_max) {
          boundary = (int)pf != k;
          if ((int)pf > k) atomicAdd(&cnt[m], 1);
        }
      }
    }
  }
  if (boundary) first[idx] = i;
  // runs inside the wave: a head lane adds the length of its run
  const int64_t prev = __shfl_up(idx, 1);
  const bool valid = idx >= 0;
  const bool head = valid && (lane == 0 || prev != idx);
  const uint64_t stop = (__ballot(head) | ~__ballot(valid)) >> 1 >> lane;     // heads / gaps above this lane
  if (head) atomicAdd(&cnt[idx], stop ? __builtin_ctzll(stop) + 1 : 64 - lane);
}

__global__ __launch_bounds__(kScanThreads) void restore_scan_kernel(int *__restrict__ cnt, int64_t m, const int *__restrict__ bad_word,
                                                                    int *__restrict__ status) {
  __shared__ int part[kScanThreads / 64 + 1];
  const int bad = bad_word[0];
  block_scan_inplace(cnt, m, part);
  if (threadIdx.x == 0 && status && bad) *status = bad;
}

// more than kScanLds counts (the 102 400 (patch, source) counts of one rank of eight at 128 images: one workgroup took 82 us
// over them, tile after tile): a workgroup per tile scans its own kScanLds counts and hands its total on; restore_scan_kernel
// then scans the totals and the scatter adds a tile's offset to the local start
__global__ __launch_bounds__(kScanThreads) void restore_scan_tiles_kernel(int *__restrict__ cnt, int64_t m, int *__restrict__ tile_sum) {
  __shared__ int part[kScanThreads / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * kScanLds;
  const int64_t mm = m - base < (int64_t)kScanLds ? m - base : (int64_t)kScanLds;
  const int total = block_scan_inplace(cnt + base, mm, part);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void restore_scatter_kernel(const float *__restrict__ rows, int cols, int key_col,
                                                              int64_t n, SegStarts sg, int world, int key_max,
                                                              const int *__restrict__ start,
                                                              const int64_t *__restrict__ first, int out_cols,
                                                              int64_t total, float *__restrict__ out,
                                                              const int *__restrict__ tile_off) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float kf = rows[i * cols + key_col];
  if (!(kf >= 0.0f && kf <= (float)key_max)) return;      // padding
  const int64_t idx = (int64_t)(int)kf * world + seg_of(sg, world, i);
  const int64_t pos = (int64_t)start[idx] + (tile_off ? (int64_t)tile_off[idx / kScanLds] : 0) + (i - first[idx]);
  if (pos < 0 || pos >= total) return;                     // (only when the promise is broken: reported in `status`)
  const float *src = rows + i * cols;
  float *dst = out + pos * out_cols;
  for (int c = 0; c < out_cols; ++c) dst[c] = src[c];
}


// ---- one tiled pass without the host (round 4) ---------------------------------------------------------------------
// obb_tiled_local_f32 / obb_tiled_merge_f32 enqueue a whole rank-side phase of BASELINE configs[4]; the kept count of
// the phase's NMS is read from its result record ON THE DEVICE (word 0), so the kernels after it are launched over
// the upper bound (all rows) and the threads beyond K leave.

// image row of kept detection i (patch_rows_kernel's arithmetic, evaluated by whoever needs the row)
struct TiledSrc {
  const float *dets; const int64_t *labels, *patch; int64_t n;
  const int64_t *keep; const void *record;           // ascending keep list + result record of the per-patch NMS
  const float *patch_xy; const int64_t *patch_image; int64_t num_patches; float rate;
};
struct TiledCaps { unsigned long long cap, segcap; unsigned int ucap; };   // what the NMS was launched with

// K of the phase's NMS as the exchange publishes it: the kept count, or -(a pair capacity that suffices) when the
// call overflowed its pair lists (every rank sees every rank's word after the all-gather of the counts and repeats
// the exchange together)
constexpr int64_t kTiledRepeat = -(1ll << 62);     // (OBB_TILED_REPEAT)
__device__ __forceinline__ int64_t tiled_k_word(const void *record, const TiledCaps &c) {
  const NmsHostInfo h = *reinterpret_cast<const NmsHostInfo *>(record);
  if ((h.route & 0xFFu) == 3u) return kTiledRepeat;          // the NMS's grid route gave up: the phase is repeated as it is
  const unsigned long long need = nms_need_capacity(h, c.cap, c.segcap, c.ucap, nullptr);
  return need ? -(int64_t)need : (int64_t)h.kept;
}

__device__ __forceinline__ int64_t tiled_patch_of(const TiledSrc &t, int64_t i, int64_t &src) {
  src = t.keep[i];
  return (src >= 0 && src < t.n) ? t.patch[src] : -1;
}
__device__ __forceinline__ float tiled_image_of(const TiledSrc &t, int64_t src, int64_t p) {
  return (src >= 0 && src < t.n && p >= 0 && p < t.num_patches) ? (float)t.patch_image[p] : -1.0f;
}
__device__ __forceinline__ void tiled_row(const TiledSrc &t, int64_t src, int64_t p, float4 &a, float4 &b) {
  a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); b = make_float4(0.0f, 0.0f, -1.0f, -1.0f);
  if (src >= 0 && src < t.n && p >= 0 && p < t.num_patches) {
    const float2 *d = reinterpret_cast<const float2 *>(t.dets + src * 6);
    const float2 d0 = d[0], d1 = d[1], d2 = d[2];
    const float2 o = reinterpret_cast<const float2 *>(t.patch_xy)[p];
    a.x = (d0.x + o.x) / t.rate; a.y = (d0.y + o.y) / t.rate;
    a.z = d1.x / t.rate; a.w = d1.y / t.rate;
    b.x = d2.x; b.y = d2.y;
    b.z = (float)t.labels[src];
    b.w = (float)t.patch_image[p];
  }
}
__device__ __forceinline__ void tiled_store_row(float *__restrict__ out, int64_t pos, const float4 &a, const float4 &b, float key) {
  float *d = out + pos * 9;                      // rows of 36 bytes: 4-byte stores
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w; d[8] = key;
}

// one rank: rows in keep order, counts = [K, 0, K]
__global__ __launch_bounds__(kMergeThreads) void tiled_rows_kernel(const TiledSrc t, const TiledCaps caps, float *__restrict__ out,
                                                                   int64_t *__restrict__ counts) {
  const int64_t K = nms_record_kept(t.record);
  const int64_t i = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (i == 0) { counts[0] = K; counts[1] = 0; counts[2] = tiled_k_word(t.record, caps); }
  if (i >= K) return;
  int64_t src;
  const int64_t p = tiled_patch_of(t, i, src);
  float4 a, b;
  tiled_row(t, src, p, a, b);
  tiled_store_row(out, i, a, b, (float)p);
}

__global__ __launch_bounds__(256) void tiled_route_count_kernel(const TiledSrc t, int world, int nblocks,
                                                                int *__restrict__ blockcnt, int *__restrict__ blockviol) {
  __shared__ int hist[kRouteMaxWorld];
  __shared__ int viol;
  const int tid = threadIdx.x;
  const int64_t K = nms_record_kept(t.record);
  if (tid < kRouteMaxWorld) hist[tid] = 0;
  if (tid == 0) viol = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRouteRows;
  int bad = 0;
#pragma unroll
  for (int u = 0; u < kRouteRows / 256; ++u) {
    const int64_t i = base + u * 256 + tid;
    if (i < K) {
      int64_t src, sp;
      const int64_t p = tiled_patch_of(t, i, src);
      atomicAdd(&hist[route_bucket(tiled_image_of(t, src, p), world)], 1);
      if (i > 0 && p < tiled_patch_of(t, i - 1, sp)) ++bad;
    }
  }
  if (bad) atomicAdd(&viol, bad);
  __syncthreads();
  if (tid < world) blockcnt[(int64_t)tid * nblocks + blockIdx.x] = hist[tid];
  if (tid == 0) blockviol[blockIdx.x] = viol;
}

__global__ __launch_bounds__(kScanThreads) void tiled_route_scan_kernel(int *__restrict__ blockcnt, int world, int nblocks,
                                                                        const int *__restrict__ blockviol,
                                                                        const void *__restrict__ record, const TiledCaps caps,
                                                                        int64_t *__restrict__ counts) {
  __shared__ int part[kScanThreads / 64 + 1];
  __shared__ int vsum;
  const int tid = threadIdx.x;
  if (tid == 0) vsum = 0;
  __syncthreads();
  int v = 0;
  for (int j = tid; j < nblocks; j += kScanThreads) v += blockviol[j];
  if (v) atomicAdd(&vsum, v);
  const int total = block_scan_inplace(blockcnt, (int64_t)world * nblocks, part);
  if (tid < world) {
    const int lo = blockcnt[(int64_t)tid * nblocks];
    const int hi = tid + 1 < world ? blockcnt[(int64_t)(tid + 1) * nblocks] : total;
    counts[tid] = hi - lo;
  }
  if (tid == 0) { counts[world] = vsum; counts[world + 1] = tiled_k_word(record, caps); }
}

__global__ __launch_bounds__(256) void tiled_route_scatter_kernel(const TiledSrc t, int world, int nblocks,
                                                                  const int *__restrict__ offs, float *__restrict__ out) {
  constexpr int kU = kRouteRows / 256;
  __shared__ int wc[kU * 4 + 1][kRouteMaxWorld];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t K = nms_record_kept(t.record);
  const int64_t base = (int64_t)blockIdx.x * kRouteRows;
  if (base >= K) return;                                  // (block-uniform)
  for (int j = tid; j < (kU * 4 + 1) * kRouteMaxWorld; j += 256) (&wc[0][0])[j] = 0;
  __syncthreads();
  int bk[kU], rk[kU];
  int64_t srcs[kU], ps[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = base + u * 256 + tid;
    const bool valid = i < K;
    srcs[u] = -1; ps[u] = -1;
    if (valid) ps[u] = tiled_patch_of(t, i, srcs[u]);
    bk[u] = valid ? route_bucket(tiled_image_of(t, srcs[u], ps[u]), world) : -1;
    rk[u] = 0;
    uint64_t rem = __ballot(valid);
    while (rem) {                                    // one round per bucket present in the wave
      const int lead = __builtin_ctzll(rem);
      const int bl = __builtin_amdgcn_readlane(bk[u], lead);
      const uint64_t m = __ballot(bk[u] == bl);
      if (bk[u] == bl) rk[u] = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == lead) wc[u * 4 + wave][bl] = __popcll(m);
      rem &= ~m;
    }
  }
  __syncthreads();
  if (tid < world) {                                 // exclusive prefix over (pass, wave) in row order, per bucket
    int run = offs[(int64_t)tid * nblocks + blockIdx.x];
    for (int j = 0; j < kU * 4; ++j) { const int v = wc[j][tid]; wc[j][tid] = run; run += v; }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = base + u * 256 + tid;
    if (i >= K) continue;
    const int64_t pos = (int64_t)wc[u * 4 + wave][bk[u]] + rk[u];
    float4 a, b;
    tiled_row(t, srcs[u], ps[u], a, b);
    tiled_store_row(out, pos, a, b, (float)ps[u]);
  }
}

// kept rows of the merged NMS: out[i, :out_cols] = rows[keep[i], :out_cols] for i < K (K from the record)
__global__ __launch_bounds__(kMergeThreads) void tiled_gather_kernel(const float *__restrict__ rows, int ld, int64_t n,
                                                                     const int64_t *__restrict__ keep,
                                                                     const void *__restrict__ record, int out_cols,
                                                                     float *__restrict__ out) {
  const int64_t K = nms_record_kept(record);
  const int64_t w = (int64_t)blockIdx.x * kMergeThreads + threadIdx.x;
  if (w >= K * out_cols) return;
  const int64_t i = w / out_cols;
  const int c = (int)(w - i * out_cols);
  const int64_t s = keep[i];
  out[w] = (s >= 0 && s < n) ? rows[s * ld + c] : 0.0f;
}

struct TiledLocalLayout { size_t keep, route, nms, nms_bytes, total; int nblocks; };
static int tiled_local_layout(int64_t n, int world, int64_t pair_capacity, TiledLocalLayout &L) {
  if (n < 0 || world < 1 || world > kRouteMaxWorld) return OBB_EINVAL;
  const size_t nn = (size_t)(n > 0 ? n : 1);
  L.nblocks = (int)((nn + kRouteRows - 1) / kRouteRows);
  size_t off = 0;
  L.keep = off; off = align_up(off + nn * sizeof(int64_t), 256);
  L.route = off; off = align_up(off + ((size_t)L.nblocks * world + L.nblocks + 16) * sizeof(int), 256);
  L.nms = off;
  L.nms_bytes = obb_nms_workspace_bytes(n, OBB_KIND_RBOX, pair_capacity);
  if (L.nms_bytes == 0) return OBB_EINVAL;
  L.total = off + L.nms_bytes;
  return OBB_OK;
}

struct TiledMergeLayout { size_t restore, restore_bytes, nms, nms_bytes, total; };
static int tiled_merge_layout(int64_t total_rows, int kind, int world, int64_t key_max, int64_t pair_capacity, TiledMergeLayout &L) {
  size_t off = 0;
  L.restore = off; L.restore_bytes = 0;
  if (key_max >= 0) {
    L.restore_bytes = obb_rows_restore_workspace_bytes(world, key_max);
    if (L.restore_bytes == 0) return OBB_EINVAL;
    off = align_up(off + L.restore_bytes, 256);
  }
  L.nms = off;
  L.nms_bytes = obb_nms_workspace_bytes(total_rows, kind, pair_capacity);
  if (L.nms_bytes == 0) return OBB_EINVAL;
  L.total = off + L.nms_bytes;
  return OBB_OK;
}

static int tiled_kind(int cols) { return cols == 8 ? OBB_KIND_RBOX : (cols == 11 ? OBB_KIND_POLY : (cols == 7 ? OBB_KIND_HBB : -1)); }

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

extern "C" int obb_patch_rows_to_image_keyed_f32(const float *dets_dev, const int64_t *labels_dev,
                                                 const int64_t *patch_index_dev, int64_t n,
                                                 const int64_t *keep_dev, int64_t k, const float *patch_xy_dev,
                                                 const int64_t *patch_image_dev, int64_t num_patches, float rate,
                                                 float *rows_dev, int64_t *patch_of_row_dev, void *stream) {
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
                     num_patches, rate, rows_dev, patch_of_row_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_patch_rows_to_image_f32(const float *dets_dev, const int64_t *labels_dev,
                                           const int64_t *patch_index_dev, int64_t n,
                                           const int64_t *keep_dev, int64_t k, const float *patch_xy_dev,
                                           const int64_t *patch_image_dev, int64_t num_patches, float rate,
                                           float *rows_dev, void *stream) {
  return obb_patch_rows_to_image_keyed_f32(dets_dev, labels_dev, patch_index_dev, n, keep_dev, k, patch_xy_dev,
                                           patch_image_dev, num_patches, rate, rows_dev, nullptr, stream);
}

extern "C" int obb_merge_rows_split_ld_f32(const float *rows_dev, int64_t n, int cols, int ld, int64_t num_classes,
                                           const float *class_thrs_dev, float default_thr, float *boxes_dev,
                                           int32_t *groups_dev, float *thrs_dev, void *stream) {
  if (n < 0) return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: negative n");
  if (cols != 7 && cols != 8 && cols != 11)
    return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: rows have 8 (theta-OBB), 11 (polygon) or 7 (horizontal) columns, got %d", cols);
  if (ld < cols) return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: row pitch %d below %d columns", ld, cols);
  if (n == 0) return OBB_OK;
  if (!rows_dev || !groups_dev) return obb::set_err(OBB_EINVAL, "obb_merge_rows_split_f32: null pointer");
  unsigned blocks;
  int rc = obb::merge_grid("obb_merge_rows_split_f32", n, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(obb::merge_split_kernel, dim3(blocks), dim3(obb::kMergeThreads), 0, (hipStream_t)stream,
                     rows_dev, cols, ld, n, num_classes, class_thrs_dev, default_thr, boxes_dev, groups_dev, thrs_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_merge_rows_split_f32(const float *rows_dev, int64_t n, int cols, int64_t num_classes,
                                        const float *class_thrs_dev, float default_thr, float *boxes_dev,
                                        int32_t *groups_dev, float *thrs_dev, void *stream) {
  return obb_merge_rows_split_ld_f32(rows_dev, n, cols, cols, num_classes, class_thrs_dev, default_thr, boxes_dev,
                                     groups_dev, thrs_dev, stream);
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

extern "C" size_t obb_rows_route_workspace_bytes(int64_t n, int world) {
  if (n < 0 || world < 1 || world > obb::kRouteMaxWorld) return 0;
  const int64_t nblocks = (n + obb::kRouteRows - 1) / obb::kRouteRows;
  return (size_t)((nblocks * world + nblocks + 16) * sizeof(int));
}

extern "C" int obb_rows_route_f32(const float *rows_dev, int64_t n, int cols, int id_col, int world,
                                  const int64_t *order_key_dev, float *out_dev, int64_t *counts_dev,
                                  void *workspace_dev, size_t workspace_bytes, void *stream) {
  if (n < 0 || cols <= 0 || id_col < 0 || id_col >= cols) return obb::set_err(OBB_EINVAL, "obb_rows_route_f32: bad size");
  if (world < 1 || world > obb::kRouteMaxWorld) return obb::set_err(OBB_EINVAL, "obb_rows_route_f32: 1 <= world <= %d", obb::kRouteMaxWorld);
  if (!counts_dev) return obb::set_err(OBB_EINVAL, "obb_rows_route_f32: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    OBB_HIP_TRY(hipMemsetAsync(counts_dev, 0, (size_t)(world + 1) * sizeof(int64_t), st));
    return OBB_OK;
  }
  if (!rows_dev || !out_dev || !workspace_dev) return obb::set_err(OBB_EINVAL, "obb_rows_route_f32: null pointer");
  if (n > (1ll << 30)) return obb::set_err(OBB_EINVAL, "obb_rows_route_f32: too many rows");
  if (workspace_bytes < obb_rows_route_workspace_bytes(n, world))
    return obb::set_err(OBB_EWORKSPACE, "obb_rows_route_f32: workspace of %zu bytes, %zu needed", workspace_bytes,
                        obb_rows_route_workspace_bytes(n, world));
  const int nblocks = (int)((n + obb::kRouteRows - 1) / obb::kRouteRows);
  int *blockcnt = static_cast<int *>(workspace_dev);
  int *blockviol = blockcnt + (size_t)nblocks * world;
  hipLaunchKernelGGL(obb::route_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, rows_dev, cols, id_col, n, world,
                     order_key_dev, nblocks, blockcnt, blockviol);
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(obb::route_scan_kernel, dim3(1), dim3(obb::kScanThreads), 0, st, blockcnt, world, nblocks, blockviol, counts_dev);
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(obb::route_scatter_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, rows_dev, cols, id_col, n, world,
                     order_key_dev, nblocks, blockcnt, out_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" size_t obb_rows_restore_workspace_bytes(int world, int64_t key_max) {
  if (world < 1 || world > obb::kRouteMaxWorld || key_max < 0) return 0;
  const int64_t m = (key_max + 1) * world;
  if (m > OBB_RESTORE_MAX_SLOTS) return 0;
  return (size_t)((m + 2) * sizeof(int) + 8 + m * sizeof(int64_t) + (size_t)(m / obb::kScanLds + 2) * sizeof(int));      // counts | first rows | tile totals
}

extern "C" int obb_rows_restore_f32(const float *rows_dev, int64_t n, int cols, int key_col, const int64_t *seg_start_host,
                                    int world, int64_t key_max, int out_cols, int64_t total, float *out_dev,
                                    int32_t *status_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
  if (n < 0 || cols <= 0 || key_col < 0 || key_col >= cols || out_cols <= 0 || out_cols > cols || total < 0 || total > n)
    return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: bad size");
  if (world < 1 || world > obb::kRouteMaxWorld || !seg_start_host) return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: 1 <= world <= %d", obb::kRouteMaxWorld);
  const size_t need = obb_rows_restore_workspace_bytes(world, key_max);
  if (key_max < 0 || key_max >= (1 << 24) || need == 0)
    return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: (key_max + 1) * world <= %d", OBB_RESTORE_MAX_SLOTS);
  if (n == 0 || total == 0) return OBB_OK;
  if (!rows_dev || !out_dev || !workspace_dev) return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: null pointer");
  if (n > (1ll << 31) - 1) return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: too many rows");
  if (workspace_bytes < need) return obb::set_err(OBB_EWORKSPACE, "obb_rows_restore_f32: workspace of %zu bytes, %zu needed", workspace_bytes, need);
  obb::SegStarts sg;
  for (int r = 0; r <= obb::kRouteMaxWorld; ++r) sg.s[r] = r < world ? seg_start_host[r] : n;
  if (sg.s[0] != 0) return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: the first source starts at row 0");
  for (int r = 1; r < world; ++r)
    if (sg.s[r] < sg.s[r - 1] || sg.s[r] > n) return obb::set_err(OBB_EINVAL, "obb_rows_restore_f32: source starts must ascend within [0, n]");
  const int64_t m = (key_max + 1) * world;
  hipStream_t st = (hipStream_t)stream;
  int *cnt = static_cast<int *>(workspace_dev);
  int64_t *first = reinterpret_cast<int64_t *>(static_cast<char *>(workspace_dev) + (((size_t)(m + 2) * sizeof(int) + 7) & ~(size_t)7));
  OBB_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)(m + 1) * sizeof(int), st));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(obb::restore_mark_kernel, dim3(blocks), dim3(256), 0, st, rows_dev, cols, key_col, n, sg, world, (int)key_max, m, cnt, first);
  OBB_LAUNCH_CHECK();
  int *tile_off = nullptr;
  if (m <= obb::kScanLds) {
    hipLaunchKernelGGL(obb::restore_scan_kernel, dim3(1), dim3(obb::kScanThreads), 0, st, cnt, m, cnt + m, status_dev);
  } else {
    const int64_t tiles = (m + obb::kScanLds - 1) / obb::kScanLds;           // (<= OBB_RESTORE_MAX_SLOTS / kScanLds: one tile of totals)
    tile_off = reinterpret_cast<int *>(first + m);
    hipLaunchKernelGGL(obb::restore_scan_tiles_kernel, dim3((unsigned)tiles), dim3(obb::kScanThreads), 0, st, cnt, m, tile_off);
    OBB_LAUNCH_CHECK();
    hipLaunchKernelGGL(obb::restore_scan_kernel, dim3(1), dim3(obb::kScanThreads), 0, st, tile_off, tiles, cnt + m, status_dev);
  }
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(obb::restore_scatter_kernel, dim3(blocks), dim3(256), 0, st, rows_dev, cols, key_col, n, sg, world, (int)key_max,
                     cnt, first, out_cols, total, out_dev, tile_off);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

// ---- fused phases of a tiled pass ------------------------------------------------------------------------------------
extern "C" size_t obb_tiled_local_workspace_bytes(int64_t n, int world, int64_t pair_capacity) {
  obb::TiledLocalLayout L;
  if (obb::tiled_local_layout(n, world, pair_capacity, L) != OBB_OK) return 0;
  return L.total;
}

extern "C" int obb_tiled_local_f32(const float *dets_dev, const int64_t *labels_dev, const int64_t *patch_index_dev,
                                   int64_t n, const float *patch_xy_dev, const int64_t *patch_image_dev,
                                   int64_t num_patches, int64_t num_classes, float rate, float iou_thr, int world,
                                   float *rows_dev, int64_t *counts_dev, void *workspace_dev, size_t workspace_bytes,
                                   void *stream) {
  using namespace obb;
  const char *who = "obb_tiled_local_f32";
  if (n < 0 || num_patches < 0 || num_classes < 1) return set_err(OBB_EINVAL, "%s: bad size", who);
  if (world < 1 || world > kRouteMaxWorld) return set_err(OBB_EINVAL, "%s: 1 <= world <= %d", who, kRouteMaxWorld);
  if (!(rate > 0.0f)) return set_err(OBB_EINVAL, "%s: rate must be positive", who);
  if (!counts_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  hipStream_t st = (hipStream_t)stream;
  const size_t cbytes = (size_t)(world + 2) * sizeof(int64_t) + OBB_NMS_RESULT_BYTES;
  if (n == 0) {      // nothing detected: zero counts, an all-zero record (K = 0)
    OBB_HIP_TRY(hipMemsetAsync(counts_dev, 0, cbytes, st));
    return OBB_OK;
  }
  if (!dets_dev || !labels_dev || !patch_index_dev || !patch_xy_dev || !patch_image_dev || !rows_dev || !workspace_dev)
    return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (n > (1ll << 30)) return set_err(OBB_EINVAL, "%s: too many rows", who);
  if (reinterpret_cast<uintptr_t>(workspace_dev) & 255u) return set_err(OBB_EINVAL, "%s: workspace must be 256-byte aligned", who);
  TiledLocalLayout L;
  if (tiled_local_layout(n, world, 1, L) != OBB_OK) return set_err(OBB_EINVAL, "%s: bad size", who);
  if (workspace_bytes <= L.nms) return set_err(OBB_EWORKSPACE, "%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, obb_tiled_local_workspace_bytes(n, world, 0));
  char *ws = static_cast<char *>(workspace_dev);
  int64_t *keep = reinterpret_cast<int64_t *>(ws + L.keep);
  void *record = counts_dev + world + 2;
  NmsSource src = {};
  src.dets = dets_dev; src.pitch = 6; src.mode = 1;
  src.a64 = labels_dev; src.b64 = patch_index_dev; src.mult = num_classes;
  src.allow_grid = 1;          // (a phase whose NMS gives up on the grid is repeated by the caller: kTiledRepeat / OBB_ERETRY)
  NmsEnqueued q;
  int rc = nms_enqueue_source(who, src, n, OBB_KIND_RBOX, iou_thr, 0, keep, record, ws + L.nms, workspace_bytes - L.nms, st, q);
  if (rc != OBB_OK) return rc;
  TiledSrc t = {dets_dev, labels_dev, patch_index_dev, n, keep, record, patch_xy_dev, patch_image_dev, num_patches, rate};
  const TiledCaps caps = {q.cap, q.segcap, q.ucap};
  if (world == 1) {
    hipLaunchKernelGGL(tiled_rows_kernel, dim3((unsigned)((n + kMergeThreads - 1) / kMergeThreads)), dim3(kMergeThreads), 0, st, t, caps, rows_dev, counts_dev);
    OBB_LAUNCH_CHECK();
    return OBB_OK;
  }
  int *blockcnt = reinterpret_cast<int *>(ws + L.route);
  int *blockviol = blockcnt + (size_t)L.nblocks * world;
  hipLaunchKernelGGL(tiled_route_count_kernel, dim3((unsigned)L.nblocks), dim3(256), 0, st, t, world, L.nblocks, blockcnt, blockviol);
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(tiled_route_scan_kernel, dim3(1), dim3(kScanThreads), 0, st, blockcnt, world, L.nblocks, blockviol, record, caps, counts_dev);
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(tiled_route_scatter_kernel, dim3((unsigned)L.nblocks), dim3(256), 0, st, t, world, L.nblocks, blockcnt, rows_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" size_t obb_tiled_merge_workspace_bytes(int64_t total_rows, int cols, int world, int64_t key_max, int64_t pair_capacity) {
  obb::TiledMergeLayout L;
  const int kind = obb::tiled_kind(cols);
  if (kind < 0 || total_rows < 0) return 0;
  if (obb::tiled_merge_layout(total_rows, kind, world, key_max, pair_capacity, L) != OBB_OK) return 0;
  return L.total;
}

extern "C" int obb_tiled_merge_f32(const float *recv_dev, int64_t m, int cols, int ld, const int64_t *seg_start_host,
                                   int world, int64_t key_max, int64_t total_rows, int64_t num_classes,
                                   const float *class_thrs_dev, float iou_thr, int flags, float *ordered_dev,
                                   float *kept_dev, int out_cols, int64_t *keep_dev, void *result_dev,
                                   int32_t *status_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
  using namespace obb;
  const char *who = "obb_tiled_merge_f32";
  const int kind = tiled_kind(cols);
  if (kind < 0) return set_err(OBB_EINVAL, "%s: rows have 8 (theta-OBB), 11 (polygon) or 7 (horizontal) columns, got %d", who, cols);
  if (m < 0 || total_rows < 0 || total_rows > m || ld < cols || out_cols < 1 || out_cols > ld || num_classes < 1)
    return set_err(OBB_EINVAL, "%s: bad size", who);
  if (!result_dev) return set_err(OBB_EINVAL, "%s: null result record", who);
  hipStream_t st = (hipStream_t)stream;
  if (total_rows == 0) {
    OBB_HIP_TRY(hipMemsetAsync(result_dev, 0, OBB_NMS_RESULT_BYTES, st));
    return OBB_OK;
  }
  if (!recv_dev || !kept_dev || !keep_dev || !workspace_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (reinterpret_cast<uintptr_t>(workspace_dev) & 255u) return set_err(OBB_EINVAL, "%s: workspace must be 256-byte aligned", who);
  TiledMergeLayout L;
  if (tiled_merge_layout(total_rows, kind, world, key_max, 1, L) != OBB_OK) return set_err(OBB_EINVAL, "%s: bad size (world, key_max)", who);
  if (workspace_bytes <= L.nms) return set_err(OBB_EWORKSPACE, "%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, obb_tiled_merge_workspace_bytes(total_rows, cols, world, key_max, 0));
  char *ws = static_cast<char *>(workspace_dev);
  const float *rows = recv_dev;
  if (key_max >= 0) {      // the sources' runs -> one run in ascending key (ld - 1 = the key column)
    if (!ordered_dev || ld < cols + 1) return set_err(OBB_EINVAL, "%s: order restoration needs a key column and an output", who);
    int rc = obb_rows_restore_f32(recv_dev, m, ld, ld - 1, seg_start_host, world, key_max, ld, total_rows, ordered_dev, status_dev,
                                  ws + L.restore, L.restore_bytes, stream);
    if (rc != OBB_OK) return rc;
    rows = ordered_dev;
  } else if (total_rows != m) {
    return set_err(OBB_EINVAL, "%s: without order restoration every row is real (total_rows == m)", who);
  }
  NmsSource src = {};
  src.dets = rows; src.pitch = ld; src.mode = 2; src.mult = num_classes; src.thr_table = class_thrs_dev;
  src.allow_grid = 1;          // (obb_tiled_result says OBB_ERETRY when the grid gave up: the caller repeats the phase)
  NmsEnqueued q;
  int rc = nms_enqueue_source(who, src, total_rows, kind, iou_thr, flags, keep_dev, result_dev, ws + L.nms, workspace_bytes - L.nms, st, q);
  if (rc != OBB_OK) return rc;
  unsigned blocks;
  rc = merge_grid(who, total_rows * out_cols, blocks);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(tiled_gather_kernel, dim3(blocks), dim3(kMergeThreads), 0, st, rows, ld, total_rows, keep_dev, result_dev, out_cols, kept_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_tiled_result(const void *record_host, int phase, int64_t n, int cols, int world, int64_t key_max,
                                size_t workspace_bytes, int64_t *info_host) {
  using namespace obb;
  size_t nms_off = 0;
  if (phase == 0) {
    TiledLocalLayout L;
    if (tiled_local_layout(n, world, 1, L) != OBB_OK) return set_err(OBB_EINVAL, "obb_tiled_result: bad size");
    nms_off = L.nms;
  } else {
    TiledMergeLayout L;
    const int kind = tiled_kind(cols);
    if (kind < 0 || tiled_merge_layout(n, kind, world, key_max, 1, L) != OBB_OK) return set_err(OBB_EINVAL, "obb_tiled_result: bad size");
    nms_off = L.nms;
  }
  if (workspace_bytes <= nms_off) return set_err(OBB_EINVAL, "obb_tiled_result: workspace smaller than the phase's fixed part");
  return obb_nms_async_result(record_host, n, workspace_bytes - nms_off, info_host);
}
