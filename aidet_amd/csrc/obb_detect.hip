// obb_detect.hip -- the detector's proposal / detection selection stages as HIP kernels (gfx950).
//
// Between a head's convolutions and its NMS the reference runs a chain of small tensor ops per FPN level --
// sigmoid, top-k, gather of anchors and deltas, delta2bbox, clamp, cat (RPNHead.get_bboxes_single,
// mmdet/models/anchor_heads/rpn_head.py:55-104; delta2bbox, mmdet/core/bbox/transforms.py:34-111) -- and after
// it a per-level cut and one more top-k (:93-103).  On a GPU that is ~40 latency-bound launches around one NMS
// call (1.27 ms of a 10 ms tile, of which the NMS kernels are ~40 us).  Here:
//
//   rpn_select_decode_kernel   one workgroup per level: exact top-`nms_pre` of the level's anchors by score
//                              (3-pass radix select on the order-preserving image of the LOGIT -- the sigmoid is
//                              monotone, so the order is the reference's -- then a bitonic sort of the <= 2048
//                              selected: descending score, equal scores by ascending anchor index), gather of
//                              the selected anchors' deltas out of the (4A, H, W) map, delta2bbox in the
//                              reference's expression order, image clamp -> (n, 5) rows + level ids, level-major
//   rows_topk_kernel           rows[keep] -> per-level cut (`proposals[:nms_post]`, :96) -> the `top` best by
//                              score, descending (:100-103), one workgroup
//
// Scores are 1 / (1 + exp(-logit)) with the device expf: within ~1e-7 of torch's sigmoid; boxes within ~1e-4 px
// of torch's delta2bbox (its exp).  Which anchors are selected and their order do not depend on that rounding
// (the keys are the logits).
#include "obb_common.h"
#include "obb_geom.h"

namespace obb {

constexpr int kSelMax = 2048;       // most rows a selection keeps (nms_pre / max_num <= 2048)
constexpr int kSelThreads = 1024;
constexpr int kSelBatch = 16;       // keys a thread has in flight in the streaming passes

// larger float <=> larger key; -0 folded into +0; NaN sorts below everything (never selected before a number)
__device__ __forceinline__ uint32_t sel_key(float f) {
  if (f != f) return 1u;            // NaN: below every number, above 0 = "excluded" (rows_topk_kernel)
  const float t = f + 0.0f;
  const uint32_t u = __float_as_uint(t);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // (>= 0x007FFFFF for -inf: never 0 or 1)
}

// In-place bitonic sort of k <= 2048 entries by (key descending, index ascending); the arrays hold kSelMax entries.
// A thread keeps the entries at positions t and t + 1024 in registers (positions >= k: padding that sorts last): the
// exchange at distance 1024 is inside the thread, distances below 64 are wave shuffles, only the 14 steps at distances
// 64..512 go through LDS and a barrier.  (The all-LDS network was 66 barrier steps, ~30 us at k = 2048, the larger part
// of every selection kernel; a rank sort -- count the entries that come before each one -- needs no barrier but k^2
// compares: 80-100 us on one CU.)
__device__ __forceinline__ void bitonic_sort(uint32_t *sk, uint32_t *si, int k) {
  static_assert(kSelMax == 2048 && kSelThreads == 1024, "two entries per thread");
  const int t = threadIdx.x;
  uint32_t k0 = t < k ? sk[t] : 0u, i0 = t < k ? si[t] : 0xFFFFFFFFu;
  uint32_t k1 = t + 1024 < k ? sk[t + 1024] : 0u, i1 = t + 1024 < k ? si[t + 1024] : 0xFFFFFFFFu;
  __syncthreads();
  // keep the better (sorts first) or the worse of (mine, partner)
  auto cx = [](uint32_t &mk, uint32_t &mi, uint32_t pk, uint32_t pi, bool take_better) {
    const bool mine_first = mk > pk || (mk == pk && mi < pi);
    if (take_better != mine_first) { mk = pk; mi = pi; }
  };
  for (int size = 2; size <= 2048; size <<= 1) {
    const bool up0 = (t & size) == 0, up1 = ((t + 1024) & size) == 0;   // run sorted "best first"?
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride == 1024) {                           // (size 2048: both entries of the thread, best first)
        const bool swap = k1 > k0 || (k1 == k0 && i1 < i0);
        if (swap) { uint32_t a = k0; k0 = k1; k1 = a; a = i0; i0 = i1; i1 = a; }
        continue;
      }
      const bool am_lo = (t & stride) == 0;
      uint32_t pk0, pi0, pk1, pi1;
      if (stride < 64) {
        pk0 = (uint32_t)__shfl_xor((int)k0, stride); pi0 = (uint32_t)__shfl_xor((int)i0, stride);
        pk1 = (uint32_t)__shfl_xor((int)k1, stride); pi1 = (uint32_t)__shfl_xor((int)i1, stride);
      } else {
        sk[t] = k0; si[t] = i0; sk[t + 1024] = k1; si[t + 1024] = i1;
        __syncthreads();
        const int q = t ^ stride;
        pk0 = sk[q]; pi0 = si[q]; pk1 = sk[q + 1024]; pi1 = si[q + 1024];
        __syncthreads();
      }
      cx(k0, i0, pk0, pi0, am_lo == up0);
      cx(k1, i1, pk1, pi1, am_lo == up1);
    }
  }
  if (t < k) { sk[t] = k0; si[t] = i0; }
  if (t + 1024 < k) { sk[t + 1024] = k1; si[t + 1024] = i1; }
  __syncthreads();
}

struct SelScratch {
  uint32_t hist[2048];
  uint32_t wsum[16];
  uint32_t prefix, krem;
  int cgt, ceq;
  int wcount[16];
  int overflow;
};

// hist[digit] += 1 for the active lanes of a wave.  Keys of a detection head cluster (logits near 0, probabilities
// near the threshold): their top digits fall into one or two bins, and 10^5 LDS atomics on ONE word serialise
// (0.3 ms for the 196 608 anchors of a stride-4 level).  The three most common digits of the wave are therefore
// added once per wave; what is left (a uniform pass: nearly everything) goes through plain atomics.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t digit, bool active, int lane) {
  uint64_t rem = __ballot(active);
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    if (rem == 0ull) break;                              // (wave-uniform)
    const int leader = __ffsll((long long)rem) - 1;
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
    const uint64_t m = __ballot(active && digit == d) & rem;
    if (lane == leader) atomicAdd(&hist[d], (uint32_t)__popcll(m));
    rem &= ~m;
  }
  if ((rem >> lane) & 1ull) atomicAdd(&hist[digit], 1u);
}

// One digit of a radix select: given the counts of the candidates' digits (bin(b), b in [0, nb)), the digit d with
// count(digits > d) < krem <= count(digits >= d) -- suffix sums over the bins; sc.prefix / sc.krem move on to it.
// All 1024 threads call it; it ends with a barrier.
template <class BinFn>
__device__ __forceinline__ void radix_digit(BinFn bin, int nb, int sh, uint32_t pre, SelScratch &sc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = nb / kSelThreads > 0 ? nb / kSelThreads : 1;     // 2 bins per thread (11 bits) or 1
  uint32_t v[2] = {0u, 0u}, tot = 0u;
  for (int u = 0; u < per; ++u) { const int b = tid * per + u; v[u] = b < nb ? bin(b) : 0u; tot += v[u]; }
  // inclusive suffix sum across threads: reverse the lane / wave order
  uint32_t inc = tot;
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t2 = (uint32_t)__shfl_down((int)inc, o); if (lane + o < 64) inc += t2; }
  if (lane == 0) sc.wsum[wave] = inc;
  __syncthreads();
  uint32_t above = inc - tot;                       // bins of later lanes of this wave
  for (int w2 = wave + 1; w2 < kSelThreads / 64; ++w2) above += sc.wsum[w2];
  // bins of this thread, from the highest down
  const uint32_t kr = sc.krem;
  __syncthreads();
  for (int u = per - 1; u >= 0; --u) {
    const int b = tid * per + u;
    if (b < nb && above < kr && kr <= above + v[u]) {   // exactly one (thread, bin) satisfies this
      sc.prefix = pre | ((uint32_t)b << sh);
      sc.krem = kr - above;
    }
    above += v[u];
  }
  __syncthreads();
}

// Exact top-k of key(i), i in [0, n): on return sk[0..k) / si[0..k) hold the k largest keys and their indices,
// sorted by (key descending, index ascending).  k <= kSelMax, k <= n.  All 1024 threads call it.
// `key` is evaluated up to 4 times per element (three histogram passes + the collection): it should be a load.
// idxmap (optional): the indices reported (and used as the tie-break of the final order) are idxmap[i] instead of i.
template <class KeyFn>
__device__ void block_topk_sorted(KeyFn key, int n, int k, uint32_t *sk, uint32_t *si, SelScratch &sc,
                                  const uint32_t *idxmap = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- radix select: the k-th largest key T, digit by digit from the top (11 + 11 + 10 bits)
  if (tid == 0) { sc.prefix = 0u; sc.krem = (uint32_t)k; }
  __syncthreads();
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  uint32_t himask = 0u;
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = shifts[pass], nb = 1 << bits[pass];
    for (int b = tid; b < nb; b += kSelThreads) sc.hist[b] = 0u;
    __syncthreads();
    const uint32_t pre = sc.prefix;
    // (whole waves iterate together: hist_add votes.  kSelBatch keys per thread are requested before the first is
    // used: one key per iteration was one exposed memory round trip per 1024 keys -- 0.3 ms for a stride-4 level)
    for (int base = 0; base < n; base += kSelBatch * kSelThreads) {
      uint32_t kk[kSelBatch];
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) { const int i = base + u * kSelThreads + tid; kk[u] = i < n ? key(i) : 0u; }
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) {
        const int i = base + u * kSelThreads + tid;
        if (base + u * kSelThreads < n)                    // (block-uniform)
          hist_add(sc.hist, (kk[u] >> sh) & (uint32_t)(nb - 1), i < n && (kk[u] & himask) == pre, lane);
      }
    }
    __syncthreads();
    radix_digit([&](int b) { return sc.hist[b]; }, nb, sh, pre, sc);
    himask |= (uint32_t)(nb - 1) << sh;
  }
  const uint32_t T = sc.prefix;
  const int need_eq = (int)sc.krem;                     // how many keys == T belong to the top k
  // ---- collect: every key > T, and the first `need_eq` keys == T in index order (chunks in order; inside a chunk
  // by ballot rank)
  if (tid == 0) { sc.cgt = 0; sc.ceq = 0; }
  __syncthreads();
  const int ngt_total = k - need_eq;
  for (int base0 = 0; base0 < n; base0 += kSelBatch * kSelThreads) {
   uint32_t kb[kSelBatch];
#pragma unroll
   for (int u = 0; u < kSelBatch; ++u) { const int i = base0 + u * kSelThreads + tid; kb[u] = i < n ? key(i) : 0u; }
#pragma unroll
   for (int u = 0; u < kSelBatch; ++u) {
    const int base = base0 + u * kSelThreads;
    if (base >= n) break;                                        // (block-uniform)
    const int i = base + tid;
    const uint32_t kk = kb[u];
    const bool gt = i < n && kk > T, eq = i < n && kk == T;
    const uint64_t mg = __ballot(gt), me = __ballot(eq);
    if (__syncthreads_or((mg | me) != 0ull) == 0) continue;    // (block-uniform) nothing to take in this chunk: the usual case
    if (lane == 0) { sc.wsum[wave] = (uint32_t)__popcll(mg) | ((uint32_t)__popcll(me) << 16); }
    __syncthreads();
    int og = sc.cgt, oe = sc.ceq, tg = 0, te = 0;
    for (int w2 = 0; w2 < kSelThreads / 64; ++w2) {
      const uint32_t v = sc.wsum[w2];
      if (w2 < wave) { og += (int)(v & 0xFFFFu); oe += (int)(v >> 16); }
      tg += (int)(v & 0xFFFFu); te += (int)(v >> 16);
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    if (gt) { const int p = og + __popcll(mg & lt); sk[p] = kk; si[p] = idxmap ? idxmap[i] : (uint32_t)i; }
    if (eq) { const int r = oe + __popcll(me & lt); if (r < need_eq) { sk[ngt_total + r] = kk; si[ngt_total + r] = idxmap ? idxmap[i] : (uint32_t)i; } }
    __syncthreads();
    if (tid == 0) { sc.cgt += tg; sc.ceq += te; }
    __syncthreads();
   }
  }
  __syncthreads();
  bitonic_sort(sk, si, k);
  __syncthreads();
}

// The same for a LARGE n (a stride-4 level has 196 608 anchors, of which 2000 are wanted): a threshold from a strided
// sample first -- the sample key whose rank says "about 2 k keys are at least this large" -- then ONE pass over all keys
// that compacts the candidates above it into per-wave LDS segments (wave-local counters: no barrier in the pass), then
// the exact selection among the <= 8192 candidates.  Too few candidates (< k) or a full segment: the plain routine.
constexpr int kCandCap = 8192, kCandPerWave = kCandCap / (kSelThreads / 64);
constexpr int kSampleN = 2048;
__device__ __forceinline__ int prefilter_rank(int k, int n) {
  int q = (int)((2.0f * (float)k / (float)n) * (float)kSampleN) + 1;
  return q < 8 ? 8 : (q > kSampleN ? kSampleN : q);
}
__device__ __forceinline__ int sample_pos(int j, int n) { return (int)(((uint64_t)(uint32_t)j * 2654435761ull) % (uint64_t)n); }
template <class KeyFn>
__device__ void block_topk_sorted_big(KeyFn key, int n, int k, uint32_t *sk, uint32_t *si, uint32_t *ck, uint32_t *ci,
                                      SelScratch &sc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // sample: kSampleN strided keys; its q-th largest, q = 2 k / n of the sample, is the prefilter threshold
  // (pseudo-random positions: a fixed stride of n / 2048 is a multiple of the anchors per location on every FPN level,
  // i.e. it would sample ONE anchor shape -- whose scores are shifted against the others' -- and miss the threshold)
  const int q = prefilter_rank(k, n);
  for (int j = tid; j < kSampleN; j += kSelThreads) ck[j] = key(sample_pos(j, n));
  __syncthreads();
  block_topk_sorted([&](int j) { return ck[j]; }, kSampleN, q, sk, si, sc);
  const uint32_t T0 = sk[q - 1];
  __syncthreads();
  if (tid == 0) sc.overflow = 0;
  int mine = 0;                                          // candidates of this wave so far (wave-uniform)
  uint32_t *wk = ck + wave * kCandPerWave, *wi = ci + wave * kCandPerWave;
  for (int base = 0; base < n; base += kSelBatch * kSelThreads) {
    uint32_t kk[kSelBatch];
#pragma unroll
    for (int u = 0; u < kSelBatch; ++u) { const int i = base + u * kSelThreads + tid; kk[u] = i < n ? key(i) : 0u; }
#pragma unroll
    for (int u = 0; u < kSelBatch; ++u) {
      const int i = base + u * kSelThreads + tid;
      const bool take = i < n && kk[u] >= T0;
      const uint64_t m = __ballot(take);
      if (m == 0ull) continue;
      const int p = mine + __popcll(m & ((1ull << lane) - 1ull));
      if (take && p < kCandPerWave) { wk[p] = kk[u]; wi[p] = (uint32_t)i; }
      mine += __popcll(m);
    }
  }
  if (lane == 0) { sc.wcount[wave] = mine < kCandPerWave ? mine : kCandPerWave; if (mine > kCandPerWave) sc.overflow = 1; }
  __syncthreads();
  int total = 0, off = 0;
  for (int w2 = 0; w2 < kSelThreads / 64; ++w2) { off += (w2 < wave) ? sc.wcount[w2] : 0; total += sc.wcount[w2]; }
  const bool fallback = sc.overflow != 0 || total < k;   // (block-uniform)
  if (fallback) {
    __syncthreads();
    block_topk_sorted(key, n, k, sk, si, sc);
    return;
  }
  // candidates -> one dense list (in place: wave w's segment moves down to offset `off`; waves go in order)
  const int cnt = sc.wcount[wave];
  uint32_t vk[kCandPerWave / 64], vi[kCandPerWave / 64];
#pragma unroll
  for (int u = 0; u < kCandPerWave / 64; ++u) { const int j = lane + u * 64; vk[u] = j < cnt ? wk[j] : 0u; vi[u] = j < cnt ? wi[j] : 0u; }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kCandPerWave / 64; ++u) { const int j = lane + u * 64; if (j < cnt) { ck[off + j] = vk[u]; ci[off + j] = vi[u]; } }
  __syncthreads();
  // exact top-k among the candidates, reported and tie-broken by ANCHOR index (which members of a tie AT the k-th key
  // are taken follows the candidate order -- wave-major -- not the anchor order: torch.topk leaves that open too)
  block_topk_sorted([&](int j) { return ck[j]; }, total, k, sk, si, sc, ci);
}

struct RpnLevels {
  const float *cls[OBB_MAX_LEVELS];      // (A * cls_out, H, W) logits
  const float *reg[OBB_MAX_LEVELS];      // (A * 4, H, W) deltas
  const float *anchors[OBB_MAX_LEVELS];  // (H * W * A, 4)
  int H[OBB_MAX_LEVELS], W[OBB_MAX_LEVELS];
  int out_base[OBB_MAX_LEVELS + 1];      // first output row of every level
};

struct BoxCodec { float m[4], s[4]; float max_ratio; int max_h, max_w; };

// delta2bbox (transforms.py:34-111) for one box, the reference's expression order
__device__ __forceinline__ float4 decode_box(float ax1, float ay1, float ax2, float ay2, float d0, float d1, float d2,
                                             float d3, const BoxCodec &c) {
  const float dx = d0 * c.s[0] + c.m[0], dy = d1 * c.s[1] + c.m[1];
  float dw = d2 * c.s[2] + c.m[2], dh = d3 * c.s[3] + c.m[3];
  dw = dw < -c.max_ratio ? -c.max_ratio : (dw > c.max_ratio ? c.max_ratio : dw);
  dh = dh < -c.max_ratio ? -c.max_ratio : (dh > c.max_ratio ? c.max_ratio : dh);
  const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f;
  const float pw = ax2 - ax1 + 1.0f, ph = ay2 - ay1 + 1.0f;
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  float x1 = gx - gw * 0.5f + 0.5f, y1 = gy - gh * 0.5f + 0.5f;
  float x2 = gx + gw * 0.5f - 0.5f, y2 = gy + gh * 0.5f - 0.5f;
  if (c.max_h > 0) {
    const float mw = (float)(c.max_w - 1), mh = (float)(c.max_h - 1);
    x1 = x1 < 0.0f ? 0.0f : (x1 > mw ? mw : x1); x2 = x2 < 0.0f ? 0.0f : (x2 > mw ? mw : x2);
    y1 = y1 < 0.0f ? 0.0f : (y1 > mh ? mh : y1); y2 = y2 < 0.0f ? 0.0f : (y2 > mh ? mh : y2);
  }
  return make_float4(x1, y1, x2, y2);
}

// ---- large levels over several workgroups (with a workspace) -----------------------------------------------------
// One workgroup walking the 196 608 anchors of a stride-4 level is 0.32 ms of a 0.8 ms RPN stage (the sampled
// prefilter above does not help where the logits cluster -- a random-init head: thousands of keys AT the threshold).
// With a workspace a level is dealt to one workgroup per SLICE of 8192 anchors and selected EXACTLY by all of them
// together: every thread keeps its 8 keys in registers (the level is read ONCE), each radix digit is a block
// histogram added to the level's global one, a barrier among the level's workgroups (they are few -- 24 for a stride-4
// level -- and co-resident: one per CU), and the same digit search in every workgroup; then the counts of keys above /
// at the threshold per slice, one more barrier, and every workgroup writes its share of the k selected (key, anchor)
// pairs -- keys above the threshold, then the first ties in ANCHOR order -- to the level's list.  The level's
// workgroup of rpn_select_decode_kernel sorts and decodes that list.
constexpr int kSliceAnchors = 8192;      // anchors per workgroup (8 per thread)
constexpr int kMaxSlices = 64;           // slices per level (levels of up to 524 288 anchors)
constexpr int kGroupMaxBlocks = 192;     // workgroups of one launch: all must be resident at once (256 CUs)
constexpr int kGroupWords = 3 * 2048 + 8;   // per level: three global histograms + barrier counters
constexpr int kGroupFailWord = 4;           // bar[0..3]: the four barriers; bar[4]: a barrier of the level timed out
struct RpnGroup {
  int first_slice[OBB_MAX_LEVELS + 1];   // first slice of every level (a level without slices: same as the next)
  uint32_t *words;                       // [levels][kGroupWords] zeroed before the launch
  uint32_t *bcnt;                        // [slices][2] keys above / at the threshold
  uint2 *sel;                            // [levels][kSelMax] the selected (key, anchor index) pairs
  int64_t spin_ticks;                    // bound of a barrier wait (wall clock, 100 MHz)
};

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// barrier among the `members` workgroups of a level (counter zero before the launch, used once)
// (what the workgroups exchange is written by agent-scope atomics only -- performed at the coherence point, nothing
// dirty in a cache -- and read back by agent-scope loads: every wave waits for its own atomics to be acknowledged,
// then ONE thread arrives; a release / acquire fence pair would cost a cache write-back and invalidate per barrier)
// The wait is BOUNDED: co-residency of a level's workgroups is sized on the host (slices <= CUs) but not
// guaranteed -- a CU mask, another process or stream on the CUs, a partitioned device.  After kGroupSpinTicks of the
// 100 MHz wall clock the waiting workgroup raises the level's `fail` word and leaves; a workgroup that gets through a
// barrier while `fail` is up leaves too, so every member ends (possibly one barrier later), and
// rpn_select_decode_kernel, which runs after this launch, selects a failed level alone (block_topk_sorted_big).
constexpr int64_t kGroupSpinTicks = 25000000;      // 0.25 s (OBB_TUNE_RPN_BARRIER_TICKS)
__device__ __forceinline__ bool group_barrier(uint32_t *counter, uint32_t members, uint32_t *fail, int *s_flag, uint64_t ticks) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 >= ticks) { ok = 0; break; }
    }
    if (!ok) __hip_atomic_fetch_or(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

__global__ __launch_bounds__(kSelThreads) void rpn_group_select_kernel(const RpnLevels lv, int A, int use_sigmoid, int num_levels,
                                                                      const RpnGroup g) {
  __shared__ SelScratch sc;
  __shared__ int wc[2][kSliceAnchors / kSelThreads][kSelThreads / 64];
  __shared__ uint32_t soff[2];
  __shared__ int s_bar_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int l = 0;
  while (l + 1 < num_levels && (int)blockIdx.x >= g.first_slice[l + 1]) ++l;
  const int slice = (int)blockIdx.x - g.first_slice[l];
  const uint32_t members = (uint32_t)(g.first_slice[l + 1] - g.first_slice[l]);
  const int HW = lv.H[l] * lv.W[l], n = HW * A;
  const float *__restrict__ cls = lv.cls[l];
  const int k = lv.out_base[l + 1] - lv.out_base[l];
  uint32_t *ghist = g.words + (size_t)l * kGroupWords, *bar = ghist + 3 * 2048;
  constexpr int kU = kSliceAnchors / kSelThreads;
  uint32_t kk[kU];
  bool in[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int i = slice * kSliceAnchors + u * kSelThreads + tid;
    in[u] = i < n;
    const int ii = in[u] ? i : n - 1;
    const int p = ii / A, a = ii - p * A;
    kk[u] = sel_key(use_sigmoid ? cls[a * HW + p] : cls[(2 * a + 1) * HW + p] - cls[(2 * a) * HW + p]);
  }
  if (tid == 0) { sc.prefix = 0u; sc.krem = (uint32_t)k; }
  __syncthreads();
  uint32_t himask = 0u;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = pass == 0 ? 21 : (pass == 1 ? 10 : 0), nb = pass == 2 ? 1024 : 2048;
    for (int b = tid; b < nb; b += kSelThreads) sc.hist[b] = 0u;
    __syncthreads();
    const uint32_t pre = sc.prefix;
#pragma unroll
    for (int u = 0; u < kU; ++u) hist_add(sc.hist, (kk[u] >> sh) & (uint32_t)(nb - 1), in[u] && (kk[u] & himask) == pre, lane);
    __syncthreads();
    uint32_t *gh = ghist + pass * 2048;
    for (int b = tid; b < nb; b += kSelThreads) { const uint32_t v = sc.hist[b]; if (v) atomicAdd(&gh[b], v); }
    if (!group_barrier(&bar[pass], members, &bar[kGroupFailWord], &s_bar_ok, (uint64_t)g.spin_ticks)) return;
    radix_digit([&](int b) { return ld_agent(&gh[b]); }, nb, sh, pre, sc);
    himask |= (uint32_t)(nb - 1) << sh;
  }
  const uint32_t T = sc.prefix;
  const int need_eq = (int)sc.krem, ngt_total = k - need_eq;
  // ---- this slice's keys above / at the threshold, in anchor order (pass-major, then thread)
  uint64_t mg[kU], me[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    mg[u] = __ballot(in[u] && kk[u] > T);
    me[u] = __ballot(in[u] && kk[u] == T);
    if (lane == 0) { wc[0][u][wave] = __popcll(mg[u]); wc[1][u][wave] = __popcll(me[u]); }
  }
  __syncthreads();
  if (tid == 0) {
    int tg = 0, te = 0;
    for (int u = 0; u < kU; ++u)
      for (int w2 = 0; w2 < kSelThreads / 64; ++w2) { tg += wc[0][u][w2]; te += wc[1][u][w2]; }
    __hip_atomic_store(&g.bcnt[2 * blockIdx.x], (uint32_t)tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&g.bcnt[2 * blockIdx.x + 1], (uint32_t)te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!group_barrier(&bar[3], members, &bar[kGroupFailWord], &s_bar_ok, (uint64_t)g.spin_ticks)) return;
  if (wave == 0) {                                     // keys of the earlier slices of this level (<= 64 slices: one wave)
    uint32_t vg = 0, ve = 0;
    if (lane < slice) { vg = ld_agent(&g.bcnt[2 * (g.first_slice[l] + lane)]); ve = ld_agent(&g.bcnt[2 * (g.first_slice[l] + lane) + 1]); }
    for (int o = 32; o > 0; o >>= 1) { vg += (uint32_t)__shfl_xor((int)vg, o); ve += (uint32_t)__shfl_xor((int)ve, o); }
    if (lane == 0) { soff[0] = vg; soff[1] = ve; }
  }
  __syncthreads();
  uint2 *out = g.sel + (size_t)l * kSelMax;
  int rung = (int)soff[0], rune = (int)soff[1];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    int og = 0, oe = 0, tg = 0, te = 0;
#pragma unroll
    for (int w2 = 0; w2 < kSelThreads / 64; ++w2) {
      const int a = wc[0][u][w2], b2 = wc[1][u][w2];
      og += (w2 < wave) ? a : 0; oe += (w2 < wave) ? b2 : 0; tg += a; te += b2;
    }
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t idx = (uint32_t)(slice * kSliceAnchors + u * kSelThreads + tid);
    if ((mg[u] >> lane) & 1ull) { const int pos = rung + og + __popcll(mg[u] & lt); if (pos < ngt_total) out[pos] = make_uint2(kk[u], idx); }
    if ((me[u] >> lane) & 1ull) { const int r = rune + oe + __popcll(me[u] & lt); if (r < need_eq) out[ngt_total + r] = make_uint2(kk[u], idx); }
    rung += tg; rune += te;
  }
}

__global__ __launch_bounds__(kSelThreads) void rpn_select_decode_kernel(const RpnLevels lv, int A, int use_sigmoid,
                                                                        int nms_pre, const BoxCodec codec,
                                                                        float *__restrict__ dets,
                                                                        int32_t *__restrict__ levels, const RpnGroup g) {
  __shared__ uint32_t sk[kSelMax], si[kSelMax];
  __shared__ uint32_t ck[kCandCap], ci[kCandCap];
  __shared__ SelScratch sc;
  const int l = blockIdx.x, tid = threadIdx.x;
  const int H = lv.H[l], W = lv.W[l], HW = H * W;
  const int n = HW * A;
  const float *__restrict__ cls = lv.cls[l];
  const float *__restrict__ reg = lv.reg[l];
  const float *__restrict__ anc = lv.anchors[l];
  const int k = lv.out_base[l + 1] - lv.out_base[l];    // min(nms_pre, n) (nms_pre <= 0: n)
  // anchor index i = p * A + a (the order of cls.permute(1, 2, 0).reshape(-1), rpn_head.py:66-75); its logit sits
  // at cls[a * HW + p] (sigmoid) or is the difference of the two class logits (softmax: p(fg) = sigmoid(l1 - l0))
  auto logit = [&](int i) -> float {
    const int p = i / A, a = i - p * A;
    if (use_sigmoid) return cls[a * HW + p];
    return cls[(2 * a + 1) * HW + p] - cls[(2 * a) * HW + p];
  };
  const bool select = k < n;
  bool done = false;
  if (select && g.sel && g.first_slice[l + 1] > g.first_slice[l] &&
      ld_agent(g.words + (size_t)l * kGroupWords + 3 * 2048 + kGroupFailWord) == 0u) {      // (block-uniform)
    // the k (key, anchor) pairs rpn_group_select_kernel left for this level: sort them
    const uint2 *list = g.sel + (size_t)l * kSelMax;
    for (int j = tid; j < k; j += kSelThreads) { const uint2 e = list[j]; sk[j] = e.x; si[j] = e.y; }
    __syncthreads();
    bitonic_sort(sk, si, k);
    done = true;
  }
  if (select && !done) {
    if (n >= 4 * kCandCap) block_topk_sorted_big([&](int i) { return sel_key(logit(i)); }, n, k, sk, si, ck, ci, sc);
    else block_topk_sorted([&](int i) { return sel_key(logit(i)); }, n, k, sk, si, sc);
  }
  float *out = dets + (size_t)lv.out_base[l] * 5;
  int32_t *olv = levels + lv.out_base[l];
  for (int j = tid; j < k; j += kSelThreads) {
    const int i = select ? (int)si[j] : j;
    const int p = i / A, a = i - p * A;
    const float4 an = reinterpret_cast<const float4 *>(anc)[i];
    const float *r = reg + (size_t)(a * 4) * HW + p;
    const float4 b = decode_box(an.x, an.y, an.z, an.w, r[0], r[HW], r[2 * HW], r[3 * HW], codec);
    const float s = 1.0f / (1.0f + expf(-logit(i)));
    out[j * 5 + 0] = b.x; out[j * 5 + 1] = b.y; out[j * 5 + 2] = b.z; out[j * 5 + 3] = b.w; out[j * 5 + 4] = s;
    olv[j] = l;
  }
}


// ---- detection heads: softmax + score threshold + decode of the (RoI, class) candidates -----------------------------
// Replaces: BBoxHead.get_det_bboxes up to its NMS (mmdet/models/bbox_heads/bbox_head.py:139-176: softmax :149,
//           delta2bbox over ALL classes :152-153, rescale :160-166) and the candidate list the class-wise NMS builds
//           (multiclass_nms_with_index, rbbox_nms.py:22-43: per class `scores[:, i] > score_thr`, boxes of that class);
//           RBBoxHead.get_det_rbboxes_parallel, rbbox_head.py:253-296 (delta2thetaobb, core/rbbox/transforms.py:356-395).
// One workgroup: probabilities of up to 2048 RoIs into LDS (one RoI per thread), then class by class an ORDERED
// compaction of the RoIs above the threshold (ballot ranks: the class-major, RoI-ascending list that nonzero() of
// the (class, RoI) mask yields), then one decode per listed candidate -- instead of decoding every class of every RoI.
constexpr int kHeadMaxRois = 2048;
constexpr int kHeadMaxClasses = 16;     // foreground classes whose probabilities fit the LDS matrix below (128 KiB)

struct HeadCodec { float m[5], s[5]; float max_ratio; int max_h, max_w; float inv_scale; int rescale; int dim; };

// dim 4: delta2bbox (transforms.py:34-111) + optional division by scale_factor; dim 5: delta2thetaobb
// (core/rbbox/transforms.py:356-395: anchor angle -pi/2, no clamp to the image) + thetaobb_rescale
__device__ __forceinline__ void head_decode(const float *__restrict__ roi, const float *__restrict__ d,
                                            const HeadCodec &c, float *__restrict__ out) {
  const float ax1 = roi[1], ay1 = roi[2], ax2 = roi[3], ay2 = roi[4];
  if (c.dim == 4) {
    BoxCodec bc;
    for (int k = 0; k < 4; ++k) { bc.m[k] = c.m[k]; bc.s[k] = c.s[k]; }
    bc.max_ratio = c.max_ratio; bc.max_h = c.max_h; bc.max_w = c.max_w;
    float4 b = decode_box(ax1, ay1, ax2, ay2, d[0], d[1], d[2], d[3], bc);
    if (c.rescale) { b.x = b.x / c.inv_scale; b.y = b.y / c.inv_scale; b.z = b.z / c.inv_scale; b.w = b.w / c.inv_scale; }
    out[0] = b.x; out[1] = b.y; out[2] = b.z; out[3] = b.w;
  } else {
    const float dx = d[0] * c.s[0] + c.m[0], dy = d[1] * c.s[1] + c.m[1];
    float dw = d[2] * c.s[2] + c.m[2], dh = d[3] * c.s[3] + c.m[3];
    const float dt = d[4] * c.s[4] + c.m[4];
    dw = dw < -c.max_ratio ? -c.max_ratio : (dw > c.max_ratio ? c.max_ratio : dw);
    dh = dh < -c.max_ratio ? -c.max_ratio : (dh > c.max_ratio ? c.max_ratio : dh);
    const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f;
    const float pw = ax2 - ax1 + 1.0f, ph = ay2 - ay1 + 1.0f;
    float gw = pw * expf(dw), gh = ph * expf(dh);
    float gx = px + pw * dx, gy = py + ph * dy;
    const float gt = dt + (-1.57079632679489661923f);
    if (c.rescale) { gx = gx / c.inv_scale; gy = gy / c.inv_scale; gw = gw / c.inv_scale; gh = gh / c.inv_scale; }   // (the angle is not scaled)
    out[0] = gx; out[1] = gy; out[2] = gw; out[3] = gh; out[4] = gt;
  }
}

// softmax of one row of (C1) logits: exp(x - max) / sum, classes summed in ascending order
__device__ __forceinline__ void row_softmax(const float *__restrict__ x, int C1, float &mx, float &sum) {
  mx = x[0];
  for (int c = 1; c < C1; ++c) mx = x[c] > mx ? x[c] : mx;
  sum = 0.0f;
  for (int c = 0; c < C1; ++c) sum += expf(x[c] - mx);
}

__global__ __launch_bounds__(1024) void head_select_decode_kernel(const float *__restrict__ rois, const float *__restrict__ cls_score,
                                                                  const float *__restrict__ pred, int n, int C1, int pred_cols,
                                                                  float score_thr, const HeadCodec codec,
                                                                  float *__restrict__ dets, int32_t *__restrict__ rows,
                                                                  int32_t *__restrict__ labels, int64_t *__restrict__ counts) {
  extern __shared__ float prob[];                  // [class - 1][n] probabilities of the foreground classes
  __shared__ int wcnt[16];
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = C1 - 1, d = codec.dim;
  for (int r = tid; r < n; r += 1024) {
    const float *x = cls_score + (size_t)r * C1;
    float mx, sum;
    row_softmax(x, C1, mx, sum);
    for (int c = 1; c < C1; ++c) prob[(size_t)(c - 1) * n + r] = expf(x[c] - mx) / sum;
  }
  if (tid == 0) s_total = 0;
  __syncthreads();
  // ---- class by class: the RoIs with prob > thr, ascending
  for (int c = 0; c < C; ++c) {
    const int before = s_total;
    int running = 0;
    for (int base = 0; base < n; base += 1024) {
      const int r = base + tid;
      const bool ok = r < n && prob[(size_t)c * n + r] > score_thr;
      const uint64_t mk = __ballot(ok);
      if (lane == 0) wcnt[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int w2 = 0; w2 < 16; ++w2) { const int v = wcnt[w2]; off += (w2 < wave) ? v : 0; tot += v; }
      if (ok) {
        const int p = before + running + off + __popcll(mk & ((1ull << lane) - 1ull));
        rows[p] = r; labels[p] = c;
      }
      running += tot;
      __syncthreads();
    }
    if (tid == 0) { counts[c] = running; s_total = before + running; }
    __syncthreads();
  }
  const int M = s_total;
  if (tid == 0) counts[C] = M;
  // ---- decode the listed candidates (their rows / labels were written by this workgroup: visible after the barrier)
  for (int p = tid; p < M; p += 1024) {
    const int r = rows[p], c = labels[p];
    const float *dl = pred + (size_t)r * pred_cols + (pred_cols > d ? (size_t)(c + 1) * d : 0);
    float *o = dets + (size_t)p * (d + 1);
    head_decode(rois + (size_t)r * 5, dl, codec, o);
    o[d] = prob[(size_t)c * n + r];
  }
}

// the (RoI, class) pairs a previous selection kept -> rows of ANOTHER head's outputs: its softmax score of that class and
// its decoded box of that class (thetaobb_nms_by_bbox_nms, rbbox_nms.py:64-119: "the rotated boxes that survive are
// those whose horizontal boxes survived")
__global__ __launch_bounds__(256) void head_gather_decode_kernel(const float *__restrict__ rois, const float *__restrict__ cls_score,
                                                                 const float *__restrict__ pred, int C1, int pred_cols,
                                                                 const int32_t *__restrict__ rows, const int32_t *__restrict__ labels,
                                                                 const int64_t *__restrict__ sel, int K, const HeadCodec codec,
                                                                 float *__restrict__ dets) {
  const int d = codec.dim;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K; j += gridDim.x * blockDim.x) {
    const int64_t p = sel ? sel[j] : j;
    const int r = rows[p], c = labels[p];
    const float *x = cls_score + (size_t)r * C1;
    float mx, sum;
    row_softmax(x, C1, mx, sum);
    const float *dl = pred + (size_t)r * pred_cols + (pred_cols > d ? (size_t)(c + 1) * d : 0);
    float *o = dets + (size_t)j * (d + 1);
    head_decode(rois + (size_t)r * 5, dl, codec, o);
    o[d] = expf(x[c + 1] - mx) / sum;
  }
}

struct LevelBases { int num; int base[OBB_MAX_LEVELS + 1]; };
constexpr int kRowsKeys = 12288;     // kept rows whose keys rows_topk_kernel holds in LDS (48 KiB; five levels x 2000 fit)

// rows[keep] -> per-level cut (rank inside the level < nms_post; keep is ascending and the rows are level-major) ->
// the `top` best by column `score_col`, descending (equal scores: the earlier row first)
__global__ __launch_bounds__(kSelThreads) void rows_topk_kernel(const float *__restrict__ rows, int cols, int score_col,
                                                                const int64_t *__restrict__ keep, int K,
                                                                const LevelBases lb, int nms_post, int top,
                                                                float *__restrict__ out, int64_t *__restrict__ out_idx,
                                                                int64_t *__restrict__ count) {
  __shared__ uint32_t sk[kSelMax], si[kSelMax];
  __shared__ uint32_t skeys[kRowsKeys];
  __shared__ SelScratch sc;
  __shared__ int before_level[OBB_MAX_LEVELS + 1];   // kept rows below the first row of every level
  __shared__ int nvalid;
  const int tid = threadIdx.x;
  if (tid <= lb.num) {                                // lower bound of base[tid] in keep
    int lo = 0, hi = K;
    const int64_t b = lb.base[tid];
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keep[mid] < b) lo = mid + 1; else hi = mid; }
    before_level[tid] = lo;
  }
  __syncthreads();
  if (tid == 0) {
    int v = 0;
    for (int l2 = 0; l2 < lb.num; ++l2) { const int c = before_level[l2 + 1] - before_level[l2]; v += c < nms_post ? c : nms_post; }
    nvalid = lb.num > 0 ? v : K;
  }
  __syncthreads();
  auto valid = [&](int j) -> bool {
    if (lb.num == 0) return true;
    int l2 = 0;
    while (l2 + 1 < lb.num && j >= before_level[l2 + 1]) ++l2;
    return j - before_level[l2] < nms_post;
  };
  const int k = nvalid < top ? nvalid : top;
  if (tid == 0) count[0] = k;
  if (k == 0) return;
  // the keys are read four times by the selection: once from memory (two dependent loads each), then from LDS
  if (K <= kRowsKeys) {
    for (int j = tid; j < K; j += kSelThreads) skeys[j] = valid(j) ? sel_key(rows[keep[j] * cols + score_col]) : 0u;
    __syncthreads();
    block_topk_sorted([&](int j) { return skeys[j]; }, K, k, sk, si, sc);
  } else {
    block_topk_sorted([&](int j) { return valid(j) ? sel_key(rows[keep[j] * cols + score_col]) : 0u; }, K, k, sk, si, sc);
  }
  for (int j = tid; j < k; j += kSelThreads) {
    const int64_t r = keep[si[j]];
    for (int c = 0; c < cols; ++c) out[(size_t)j * cols + c] = rows[r * cols + c];
    if (out_idx) out_idx[j] = r;
  }
}

}  // namespace obb

namespace obb {
// slices of the levels that are selected by several workgroups: a real cut (n > nms_pre) of at least two slices
static int rpn_slices(const int *H_host, const int *W_host, int num_levels, int num_anchors, int nms_pre, int *first_slice) {
  int total = 0;
  for (int l = 0; l < num_levels; ++l) {
    first_slice[l] = total;
    const int64_t n = (int64_t)H_host[l] * W_host[l] * num_anchors;
    const int64_t sl = (n + kSliceAnchors - 1) / kSliceAnchors;
    if (nms_pre > 0 && n > nms_pre && sl >= 2 && sl <= kMaxSlices && total + sl <= kGroupMaxBlocks) total += (int)sl;
  }
  first_slice[num_levels] = total;
  return total;
}
static size_t rpn_ws_zeroed(int num_levels) { return (size_t)num_levels * kGroupWords * sizeof(uint32_t); }
// the workgroups of rpn_group_select_kernel wait for each other: all of them must be resident at once, one per CU
// (1024 threads).  A partitioned device (CPX: 32 CUs) may have fewer CUs than a launch wants -> the one-workgroup path.
static int device_cus() {
  static thread_local int cached_dev = -1, cached_cus = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
    cached_dev = dev; cached_cus = v;
  }
  return cached_cus;
}
}  // namespace obb

extern "C" size_t obb_rpn_select_workspace_bytes(const int *H_host, const int *W_host, int num_levels, int num_anchors,
                                                 int nms_pre) {
  if (!H_host || !W_host || num_levels < 1 || num_levels > OBB_MAX_LEVELS || num_anchors < 1) return 0;
  int fs[OBB_MAX_LEVELS + 1];
  for (int l = 0; l < num_levels; ++l)
    if (H_host[l] < 0 || W_host[l] < 0 || (int64_t)H_host[l] * W_host[l] * num_anchors > 2147483647LL / 8) return 0;
  const int slices = obb::rpn_slices(H_host, W_host, num_levels, num_anchors, nms_pre, fs);
  if (!slices) return 0;
  return 256 + obb::rpn_ws_zeroed(num_levels) + (size_t)slices * 2 * sizeof(uint32_t) + (size_t)num_levels * obb::kSelMax * sizeof(uint2);
}

extern "C" int obb_rpn_select_decode_f32(const float *const *cls_devs_host, const float *const *reg_devs_host,
                                         const float *const *anchor_devs_host, const int *H_host, const int *W_host,
                                         int num_levels, int num_anchors, int use_sigmoid, int nms_pre,
                                         const float *means_host, const float *stds_host, float wh_ratio_clip,
                                         int max_h, int max_w, float *dets_dev, int32_t *levels_dev,
                                         int64_t *num_rows_host, void *stream) {
  return obb_rpn_select_decode_ws_f32(cls_devs_host, reg_devs_host, anchor_devs_host, H_host, W_host, num_levels, num_anchors,
                                      use_sigmoid, nms_pre, means_host, stds_host, wh_ratio_clip, max_h, max_w, dets_dev,
                                      levels_dev, num_rows_host, nullptr, 0, stream);
}

extern "C" int obb_rpn_select_decode_ws_f32(const float *const *cls_devs_host, const float *const *reg_devs_host,
                                            const float *const *anchor_devs_host, const int *H_host, const int *W_host,
                                            int num_levels, int num_anchors, int use_sigmoid, int nms_pre,
                                            const float *means_host, const float *stds_host, float wh_ratio_clip,
                                            int max_h, int max_w, float *dets_dev, int32_t *levels_dev,
                                            int64_t *num_rows_host, void *workspace_dev, size_t workspace_bytes,
                                            void *stream) {
  using namespace obb;
  const char *who = "obb_rpn_select_decode_f32";
  if (num_rows_host) *num_rows_host = 0;
  if (num_levels < 1 || num_levels > OBB_MAX_LEVELS) return set_err(OBB_EINVAL, "%s: num_levels = %d (1..%d)", who, num_levels, OBB_MAX_LEVELS);
  if (num_anchors < 1) return set_err(OBB_EINVAL, "%s: num_anchors = %d", who, num_anchors);
  if (!cls_devs_host || !reg_devs_host || !anchor_devs_host || !H_host || !W_host || !means_host || !stds_host || !num_rows_host)
    return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (nms_pre > kSelMax) return set_err(OBB_EINVAL, "%s: nms_pre = %d exceeds %d", who, nms_pre, kSelMax);
  if (!(wh_ratio_clip > 0.0f)) return set_err(OBB_EINVAL, "%s: wh_ratio_clip must be positive", who);
  RpnLevels lv = {};
  int64_t total = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (H_host[l] < 0 || W_host[l] < 0) return set_err(OBB_EINVAL, "%s: negative map size", who);
    const int64_t n = (int64_t)H_host[l] * W_host[l] * num_anchors;
    if (n > 2147483647LL / 8) return set_err(OBB_EINVAL, "%s: level %d too large", who, l);
    if (n > 0 && (!cls_devs_host[l] || !reg_devs_host[l] || !anchor_devs_host[l])) return set_err(OBB_EINVAL, "%s: null level %d", who, l);
    if (nms_pre <= 0 && n > kSelMax * 64) return set_err(OBB_EINVAL, "%s: nms_pre <= 0 with %lld anchors on level %d", who, (long long)n, l);
    lv.cls[l] = cls_devs_host[l]; lv.reg[l] = reg_devs_host[l]; lv.anchors[l] = anchor_devs_host[l];
    lv.H[l] = H_host[l]; lv.W[l] = W_host[l];
    lv.out_base[l] = (int)total;
    total += (nms_pre > 0 && n > nms_pre) ? nms_pre : n;
  }
  lv.out_base[num_levels] = (int)total;
  *num_rows_host = total;
  if (total == 0 || (!dets_dev && !levels_dev)) return OBB_OK;     // (sizing call: the row count depends on shapes only)
  if (!dets_dev || !levels_dev) return set_err(OBB_EINVAL, "%s: null output", who);
  for (int l = 0; l < num_levels; ++l)
    if (reinterpret_cast<uintptr_t>(anchor_devs_host[l]) & 15u) return set_err(OBB_EINVAL, "%s: anchors must be 16-byte aligned", who);
  BoxCodec c;
  for (int k = 0; k < 4; ++k) { c.m[k] = means_host[k]; c.s[k] = stds_host[k]; }
  c.max_ratio = fabsf(logf(wh_ratio_clip));
  c.max_h = max_h; c.max_w = max_w;
  RpnGroup g = {};
  if (workspace_dev) {
    const int slices = rpn_slices(H_host, W_host, num_levels, num_anchors, nms_pre, g.first_slice);
    if (slices > 0 && slices <= device_cus()) {
      const size_t need = obb_rpn_select_workspace_bytes(H_host, W_host, num_levels, num_anchors, nms_pre);
      if (workspace_bytes < need) return set_err(OBB_EWORKSPACE, "%s: workspace of %zu bytes, %zu needed", who, workspace_bytes, need);
      char *w = static_cast<char *>(workspace_dev);
      w += (256 - (reinterpret_cast<uintptr_t>(w) & 255u)) & 255u;
      g.words = reinterpret_cast<uint32_t *>(w);
      g.sel = reinterpret_cast<uint2 *>(w + rpn_ws_zeroed(num_levels));
      g.bcnt = reinterpret_cast<uint32_t *>(w + rpn_ws_zeroed(num_levels) + (size_t)num_levels * kSelMax * sizeof(uint2));
      g.spin_ticks = tuning(OBB_TUNE_RPN_BARRIER_TICKS, kGroupSpinTicks);
      OBB_HIP_TRY(hipMemsetAsync(g.words, 0, rpn_ws_zeroed(num_levels), (hipStream_t)stream));
      hipLaunchKernelGGL(rpn_group_select_kernel, dim3((unsigned)slices), dim3(kSelThreads), 0, (hipStream_t)stream, lv, num_anchors,
                         use_sigmoid ? 1 : 0, num_levels, g);
      OBB_LAUNCH_CHECK();
    } else {
      g = RpnGroup{};
    }
  }
  hipLaunchKernelGGL(rpn_select_decode_kernel, dim3((unsigned)num_levels), dim3(kSelThreads), 0, (hipStream_t)stream, lv, num_anchors,
                     use_sigmoid ? 1 : 0, nms_pre, c, dets_dev, levels_dev, g);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_rows_topk_f32(const float *rows_dev, int64_t num_rows, int cols, int score_col,
                                 const int64_t *keep_dev, int64_t K, const int *level_base_host, int num_levels,
                                 int nms_post, int top, float *out_dev, int64_t *out_idx_dev, int64_t *count_dev,
                                 void *stream) {
  using namespace obb;
  const char *who = "obb_rows_topk_f32";
  if (num_rows < 0 || K < 0 || cols < 1 || score_col < 0 || score_col >= cols) return set_err(OBB_EINVAL, "%s: bad shape", who);
  if (num_levels < 0 || num_levels > OBB_MAX_LEVELS) return set_err(OBB_EINVAL, "%s: num_levels = %d", who, num_levels);
  if (top < 0 || top > kSelMax) return set_err(OBB_EINVAL, "%s: top = %d (0..%d)", who, top, kSelMax);
  if (K > 2147483647LL / 4 || num_rows > 2147483647LL) return set_err(OBB_EINVAL, "%s: too many rows", who);
  if (!count_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  hipStream_t st = (hipStream_t)stream;
  if (K == 0 || top == 0) {
    OBB_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), st));
    return OBB_OK;
  }
  if (!rows_dev || !keep_dev || !out_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  LevelBases lb = {};
  lb.num = num_levels;
  if (num_levels > 0) {
    if (!level_base_host) return set_err(OBB_EINVAL, "%s: null level bases", who);
    for (int l = 0; l <= num_levels; ++l) lb.base[l] = level_base_host[l];
  }
  if (nms_post <= 0) nms_post = 2147483647;
  hipLaunchKernelGGL(rows_topk_kernel, dim3(1), dim3(kSelThreads), 0, st, rows_dev, cols, score_col, keep_dev, (int)K, lb, nms_post, top,
                     out_dev, out_idx_dev, count_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

namespace obb {
static int head_codec(const char *who, int dim, const float *means_host, const float *stds_host, float wh_ratio_clip, int max_h,
                      int max_w, float scale_factor, HeadCodec &c) {
  if (dim != 4 && dim != 5) return set_err(OBB_EINVAL, "%s: dim = %d (4 horizontal, 5 theta-OBB)", who, dim);
  if (!means_host || !stds_host) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (!(wh_ratio_clip > 0.0f)) return set_err(OBB_EINVAL, "%s: wh_ratio_clip must be positive", who);
  for (int k = 0; k < 5; ++k) { c.m[k] = k < dim ? means_host[k] : 0.0f; c.s[k] = k < dim ? stds_host[k] : 1.0f; }
  c.max_ratio = fabsf(logf(wh_ratio_clip));
  c.max_h = max_h; c.max_w = max_w;
  c.rescale = scale_factor > 0.0f ? 1 : 0;
  c.inv_scale = scale_factor > 0.0f ? scale_factor : 1.0f;
  c.dim = dim;
  return OBB_OK;
}
}  // namespace obb

extern "C" int obb_head_select_decode_f32(const float *rois_dev, const float *cls_score_dev, const float *pred_dev,
                                          int64_t n, int num_classes, int pred_cols, int dim, float score_thr,
                                          const float *means_host, const float *stds_host, float wh_ratio_clip,
                                          int max_h, int max_w, float scale_factor, float *dets_dev,
                                          int32_t *rows_dev, int32_t *labels_dev, int64_t *counts_dev, void *stream) {
  using namespace obb;
  const char *who = "obb_head_select_decode_f32";
  if (n < 0 || num_classes < 2) return set_err(OBB_EINVAL, "%s: n = %lld, num_classes = %d", who, (long long)n, num_classes);
  if (n > kHeadMaxRois || num_classes - 1 > kHeadMaxClasses)
    return set_err(OBB_EINVAL, "%s: at most %d RoIs and %d foreground classes", who, kHeadMaxRois, kHeadMaxClasses);
  if (pred_cols != dim && pred_cols != dim * num_classes) return set_err(OBB_EINVAL, "%s: pred_cols = %d (dim or dim * num_classes)", who, pred_cols);
  HeadCodec c;
  int rc = head_codec(who, dim, means_host, stds_host, wh_ratio_clip, max_h, max_w, scale_factor, c);
  if (rc != OBB_OK) return rc;
  if (!counts_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    OBB_HIP_TRY(hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * (size_t)num_classes, st));
    return OBB_OK;
  }
  if (!rois_dev || !cls_score_dev || !pred_dev || !dets_dev || !rows_dev || !labels_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  const size_t lds = (size_t)(num_classes - 1) * (size_t)n * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {   // (up to 128 KiB of dynamic LDS: above the 64 KiB a kernel gets without asking)
    OBB_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(head_select_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kHeadMaxRois * kHeadMaxClasses * (int)sizeof(float)));
    attr_set = true;
  }
  hipLaunchKernelGGL(head_select_decode_kernel, dim3(1), dim3(1024), lds, st, rois_dev, cls_score_dev, pred_dev, (int)n, num_classes,
                     pred_cols, score_thr, c, dets_dev, rows_dev, labels_dev, counts_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_head_gather_decode_f32(const float *rois_dev, const float *cls_score_dev, const float *pred_dev,
                                          int num_classes, int pred_cols, int dim, const int32_t *rows_dev,
                                          const int32_t *labels_dev, const int64_t *sel_dev, int64_t K,
                                          const float *means_host, const float *stds_host, float wh_ratio_clip,
                                          int max_h, int max_w, float scale_factor, float *dets_dev, void *stream) {
  using namespace obb;
  const char *who = "obb_head_gather_decode_f32";
  if (K < 0 || num_classes < 2 || K > 2147483647LL) return set_err(OBB_EINVAL, "%s: bad size", who);
  if (pred_cols != dim && pred_cols != dim * num_classes) return set_err(OBB_EINVAL, "%s: pred_cols = %d (dim or dim * num_classes)", who, pred_cols);
  HeadCodec c;
  int rc = head_codec(who, dim, means_host, stds_host, wh_ratio_clip, max_h, max_w, scale_factor, c);
  if (rc != OBB_OK) return rc;
  if (K == 0) return OBB_OK;
  if (!rois_dev || !cls_score_dev || !pred_dev || !rows_dev || !labels_dev || !dets_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  const unsigned blocks = (unsigned)((K + 255) / 256);
  hipLaunchKernelGGL(head_gather_decode_kernel, dim3(blocks > 1024u ? 1024u : blocks), dim3(256), 0, (hipStream_t)stream, rois_dev, cls_score_dev,
                     pred_dev, num_classes, pred_cols, rows_dev, labels_dev, sel_dev, (int)K, c, dets_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
