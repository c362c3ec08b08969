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
#include "obb_geom.h"
#include "obb_rroi_internal.h"

namespace obb {

constexpr int kChanPerThread = 8;   // channel grouping of the direct (unstaged) paths


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
constexpr int kStageFloats = 5120;    // backward: most floats of one staging pass ([channel][odd pitch]; see kBwdStageFloats)
constexpr int kMaxBins = 64;          // bins of a staged RoI
constexpr int kMaxPassChan = 16;      // channels per staging pass (gradient tile 16 x 65 floats)
constexpr int kBinPitchMax = 65;


// XCD-aware block -> (RoI, channel chunk): consecutive ids take the SAME RoI and 8 DIFFERENT
// chunks (one per XCD under round-robin dispatch); chunks 8r .. 8r+7 ("round" r) are swept
// over all RoIs before round r + 1 starts.  A last round of fewer than 8 chunks (C = 128: four chunks of 32
// channels) shares its t chunks among 8 / t RoIs per group of ids, so that no XCD goes without work
// (with ids of missing chunks simply leaving, four chunks ran on half of the XCDs: twice the time).
__device__ __forceinline__ bool block_to_work(int K, int chunks, int64_t &k, int &chunk) {
  const uint32_t lin = blockIdx.x;
  const uint32_t full = (uint32_t)chunks >> 3, tail = (uint32_t)chunks & 7u;
  const uint32_t nfull = full * (uint32_t)K * 8u;
  if (lin < nfull) {
    const uint32_t c8 = lin & 7u, rest = lin >> 3;
    k = rest % (uint32_t)K;
    chunk = (int)((rest / (uint32_t)K) * 8u + c8);
    return true;
  }
  if (tail == 0u) return false;
  const uint32_t t = lin - nfull, c8 = t & 7u, rest = t >> 3;
  const uint32_t reps = 8u / tail;                 // RoIs per group of 8 ids
  if (c8 >= tail * reps) return false;
  k = (int64_t)rest * reps + c8 / tail;
  chunk = (int)(full * 8u + c8 % tail);
  return k < K;
}

// channels per staging pass: the largest power of two <= kMaxPassChan whose planes fit the
// tile (odd pitch per plane), not (much) more than the block owns
__device__ __forceinline__ int pass_channels(int cpitch, int owned) {
  int cb = kMaxPassChan;
  while (cb > 1 && (cpitch * cb > kStageFloats || (cb >> 1) >= owned)) cb >>= 1;
  return cb;
}

// sample table: footprint-local tap offsets and weights, once per block
template <int THREADS = 256>
__device__ __forceinline__ void build_table(const RoiFrame &f, const Footprint &fp, int H, int W, int pw,
                                            int spb, int nsamp, int4 *s_off, float4 *s_wgt) {
  for (int sidx = threadIdx.x; sidx < nsamp; sidx += THREADS) {
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
    s_off[sidx] = live ? make_int4(ly0 + lx0, ly0 + lx1, ly1 + lx0, ly1 + lx1) : make_int4(0, 0, 0, 0);
    s_wgt[sidx] = make_float4(w[0], w[1], w[2], w[3]);
  }
}

// ---- forward, staged ---------------------------------------------------------------------
// grid: (K x channel chunks, XCD-aware order); block: 256 threads; item = (channel group, bin).
// The RoI's footprint (the axis-aligned pixel box that contains every bilinear tap of every
// sample) of each channel plane is first copied into LDS by LDS-DMA (16 bytes per lane where
// the rows allow it: 1 KiB per wave-instruction, nothing passes through VGPRs); the ~4 x
// samples x channels taps are then LDS gathers instead of scattered 4-byte global loads.
// Values and operation order are unchanged.
// (A variant with a transposed tile -- lane = (bin, channel), bank-conflict free, every lane
// busy -- was built and measured this round: 71 us against 50 us for this one at configs[2].
// It needs dword DMA or register staging for the odd pitch, i.e. 4x the staging instructions,
// and a table read per tap group)
// Instrumented builds for the floor derivation of DESIGN.md 3.3 (`make EXTRA=-DOBB_RROI_ABLATE=n OUT=...`; results are
// WRONG on purpose, only the time is read): 1 = no sample-table arithmetic (what a per-RoI plan pass would take out of
// this kernel), 3 = one LDS read per output instead of the 16 taps (the gather loop's share), 4 = no footprint copy (the
// memory side), 5 = phase stamps (below), 6 = no output stores.
#ifndef OBB_RROI_ABLATE
#define OBB_RROI_ABLATE 0
#endif
constexpr int kAblate = OBB_RROI_ABLATE;
// OBB_RROI_ABLATE = 5: phase stamps (s_memtime, the shader clock) of every block's wave 0 -- entry | frame + footprint
// + copy offsets done, first copy issued | table built, barrier passed | first copy landed | last store issued -- read
// back with obb_debug_rroi_stamps (declared in tools/exp_rroi_stamps.py only: not part of the ABI).
#if OBB_RROI_ABLATE == 5
__device__ unsigned long long g_rroi_stamps[8 * 8192];
#define OBB_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_rroi_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define OBB_STAMP(i) do { } while (0)
#endif
constexpr int kFwdStageFloats = 5120;               // LDS tile per block (20 KiB; with the pads and the table 30 KiB: 5 blocks per CU;
                                                    // 4096 / 6 blocks per CU was measured: 39 -> 44 us at configs[2], more passes and 6 spilled registers)
constexpr int kWaveFloats = kFwdStageFloats / 4;    // private quarter of a wave (small footprints)
constexpr int kWavePad = 256;                       // room for the tail lanes of a wave's last DMA instruction

// ---- forward plan (round 4; with a workspace) ---------------------------------------------------------------------
// Phase stamps of an instrumented build (tools/exp_rroi_stamps.py) put 40 % of a forward block's life into its prologue
// -- RoI row, trigonometry, bin sizes, the four corner samples of the footprint, the copy's reciprocals: a chain of
// ~400 DEPENDENT instructions -- and 8 % into the sample table + its barrier, all of it recomputed by each of the
// C / 32 blocks of a RoI.  With a workspace one wave per RoI computes it ONCE (lane = bin: its <= 4 samples) into a
// plan record; a forward block then reads the record's header with scalar loads and every lane its own table entries
// straight into registers: no trigonometry, no LDS table, no block barrier -- the four waves of a block are independent.
// Only RoIs on the small-footprint path (state 1) are planned; everything else runs as before.
constexpr int kPlanTabEntries = 4 * 64;      // [sample of the bin][bin]
struct alignas(16) FwdPlanHdr {
  RoiFrame f;
  Footprint fp;                               // (already widened to 16-byte chunks when the copy is 16 bytes wide)
  int pitch, fsz;
  uint32_t m_slots, m_rslots;
  int state;                                  // 1: small-footprint path, table below; anything else: the block works it out itself
  int lvl, spb, pad;
};
constexpr size_t kPlanHdrBytes = 128;
static_assert(sizeof(FwdPlanHdr) <= kPlanHdrBytes, "plan header");
constexpr size_t kPlanStride = kPlanHdrBytes + kPlanTabEntries * (sizeof(ushort4) + sizeof(float4));   // 6272 bytes per RoI
constexpr int kPlanMinC = 64;                 // fewer channels: at most one block per RoI would read the record
// Measured (tools/exp_rroi_plan.py, rocprof per kernel): the planned forward kernel is NOT faster at configs[2] (42.6 vs
// 43.2 us: with the prologue gone the kernel sits on its second floor, ~190 MB of 64-byte sector traffic at ~5 TB/s) and
// the plan launch costs 7 us + a boundary: 48.3 vs 42.6 us per call.  From ~1000 RoIs on -- where the Morton order has
// already cut the traffic and a pre-launch exists anyway -- it pays: config-4 shape 168 vs 182 us for the kernel, 182 vs
// 189 us per call.  Hence the same threshold as the order launch.
constexpr int64_t kPlanMinK = 1024;
constexpr int64_t kPlanMaxK = 16384;          // (105 MB of records)

template <bool X4>
__global__ __launch_bounds__(256) void rroi_fwd_plan_kernel(const LevelTable lv, const int *__restrict__ roi_level, int num_levels,
                                                            int B, int C, const float *__restrict__ rois, int ph, int pw,
                                                            int sample_num, float min_size, int K, int roi_kind,
                                                            unsigned char *__restrict__ plan) {
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int lvl_raw = roi_level ? roi_level[k] : 0;
  const bool lvl_ok = lvl_raw >= 0 && lvl_raw < num_levels;
  const int lvl = lvl_ok ? lvl_raw : 0;
  const int H = lv.H[lvl], W = lv.W[lvl];
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, lv.scale[lvl], ph, pw, sample_num, min_size, f);
  const int nb = ph * pw;
  const long long spb64 = (long long)f.gh * f.gw;
  const bool live = lvl_ok && f.b >= 0 && f.b < B && spb64 > 0;
  Footprint fp = {0, 0, 1, 1};
  if (live) fp = roi_footprint(f, H, W, ph, pw);
  if (X4) {
    const int x1 = (fp.x0 + fp.w - 1) | 3;
    fp.x0 &= ~3;
    fp.w = x1 - fp.x0 + 1;
  }
  const int pitch = X4 ? (((fp.w >> 2) | 1) << 2) : (fp.w | 1);
  const int fsz = pitch * fp.h;
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : (spb64 > 0 ? spb64 : 1));
  const int nsamp = nb * spb;
  const bool staged = live && fsz <= kFwdStageFloats && nsamp <= kMaxTabSamples && (int64_t)B * C * ((int64_t)H * W) < (1ll << 31);
  const bool small = staged && fsz <= kWaveFloats && nb <= 64 && spb <= 4;
  unsigned char *rec = plan + (size_t)k * kPlanStride;
  if (lane == 0) {
    FwdPlanHdr h;
    h.f = f; h.fp = fp; h.pitch = pitch; h.fsz = fsz;
    const int slots = X4 ? (fsz >> 2) : fsz, rslots = X4 ? (pitch >> 2) : pitch;
    h.m_slots = slots > 1 ? 0xFFFFFFFFu / (uint32_t)slots + 1u : 0u;
    h.m_rslots = rslots > 1 ? 0xFFFFFFFFu / (uint32_t)rslots + 1u : 0u;
    h.state = small ? 1 : 0; h.lvl = lvl; h.spb = spb; h.pad = 0;
    *reinterpret_cast<FwdPlanHdr *>(rec) = h;
  }
  if (!small) return;
  ushort4 *t_off = reinterpret_cast<ushort4 *>(rec + kPlanHdrBytes);
  float4 *t_wgt = reinterpret_cast<float4 *>(rec + kPlanHdrBytes + kPlanTabEntries * sizeof(ushort4));
  if (lane < nb) {
    const int bi = lane / pw, bj = lane - bi * pw;
    for (int q = 0; q < spb; ++q) {              // the table loop of rroi_forward_kernel, sample (bin = lane, q)
      const int iy = q / f.gw, ix = q - iy * f.gw;
      float x, y, w[4];
      int yl, xl, yh, xh;
      sample_xy(f, bi, bj, iy, ix, x, y);
      const bool on = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
      auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
      const int ly0 = cl(yl - fp.y0, fp.h) * pitch, ly1 = cl(yh - fp.y0, fp.h) * pitch;
      const int lx0 = cl(xl - fp.x0, fp.w), lx1 = cl(xh - fp.x0, fp.w);
      t_off[q * 64 + lane] = on ? make_ushort4((unsigned short)(ly0 + lx0), (unsigned short)(ly0 + lx1), (unsigned short)(ly1 + lx0), (unsigned short)(ly1 + lx1))
                                : make_ushort4(0, 0, 0, 0);
      t_wgt[q * 64 + lane] = make_float4(w[0], w[1], w[2], w[3]);
    }
  }
}

template <bool X4>
__global__ __launch_bounds__(256, 5) void rroi_forward_kernel(
    const LevelTable lv, const int *__restrict__ roi_level, int num_levels, int B, int C,
    const float *__restrict__ rois, int ph, int pw, int sample_num, float min_size, int chan_per_block,
    int K, int chunks, int roi_kind, const int *__restrict__ roi_order, float *__restrict__ out,
    const unsigned char *__restrict__ plan) {
  __shared__ __attribute__((aligned(16))) float tile[4 * (kWaveFloats + kWavePad)];  // block-wide: kFwdStageFloats + pad
  __shared__ ushort4 s_off[kMaxTabSamples];   // per sample: 4 footprint-local tap offsets (< kFwdStageFloats)
  __shared__ float4 s_wgt[kMaxTabSamples];    // per sample: 4 bilinear weights
  int64_t k;
  int chunk;
  OBB_STAMP(0);
  if (!block_to_work(K, chunks, k, chunk)) return;
  // the blocks visit the RoIs in the order of `roi_order` (Morton order of their centres, rroi_order_kernel):
  // RoIs that run at the same time on an XCD then share footprint lines in its L2 (fetch 189 -> 70 MB at
  // configs[2]); where a RoI's results go does not depend on it
  if (roi_order) {
    const int kk = roi_order[k];
    if (kk >= 0 && kk < K) k = kk;
  }
  // a plan record of this RoI (rroi_fwd_plan_kernel; block-uniform: scalar loads)?
  const FwdPlanHdr *__restrict__ P = plan ? reinterpret_cast<const FwdPlanHdr *>(plan + (size_t)k * kPlanStride) : nullptr;
  const bool planned = P != nullptr && P->state == 1;
  // pyramid level of this RoI (block-uniform): feature map, its size and scale
  const int lvl_raw = planned ? P->lvl : (roi_level ? roi_level[k] : 0);
  const bool lvl_ok = lvl_raw >= 0 && lvl_raw < num_levels;
  const int lvl = lvl_ok ? lvl_raw : 0;
  const float *__restrict__ feat = lv.feat[lvl];
  const int H = lv.H[lvl], W = lv.W[lvl];
  const float scale = lv.scale[lvl];
  // The RoI's frame (trigonometry, bin sizes) and footprint (four corner samples) are the same for the whole block: ONE
  // wave computes them and hands them over through LDS.  (Round 4, from the phase stamps of an instrumented build --
  // tools/exp_rroi_stamps.py: this prologue was 40 % of a block's life, 7 900 of 19 900 cycles, because all four waves
  // executed its ~500 wave-wide VALU instructions on SIMDs shared by five waves each; the other three waves now wait
  // at the barrier and leave their issue slots to the other blocks' gathers.)
  __shared__ RoiFrame s_frame;
  __shared__ Footprint s_fp;
  __shared__ uint32_t s_magic[2];
  const int nb = ph * pw;
  const int c_block = chunk * chan_per_block;
  const int c_end = min(C, c_block + chan_per_block);
  const int64_t plane = (int64_t)H * W;
  RoiFrame f;
  Footprint fp;
  uint32_t m_slots, m_rslots;
  if (planned) {
    f = P->f; fp = P->fp; m_slots = P->m_slots; m_rslots = P->m_rslots;
  } else {
  if (threadIdx.x < 64) {
    RoiFrame f0;
    make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f0);
    Footprint fp0 = {0, 0, 1, 1};
    if (lvl_ok && f0.b >= 0 && f0.b < B && (long long)f0.gh * f0.gw > 0) fp0 = roi_footprint(f0, H, W, ph, pw);
    if (X4) {  // 16-byte LDS-DMA: rows start and end on multiples of 4 floats (W % 4 == 0)
      const int x1 = (fp0.x0 + fp0.w - 1) | 3;
      fp0.x0 &= ~3;
      fp0.w = x1 - fp0.x0 + 1;
    }
    // the two reciprocals of the copy's slot -> (channel, row, chunk) map (32-bit divisions: ~30 instructions each)
    const int pitch0 = X4 ? (((fp0.w >> 2) | 1) << 2) : (fp0.w | 1);
    const int fsz0 = pitch0 * fp0.h;
    const int slots0 = X4 ? (fsz0 >> 2) : fsz0, rslots0 = X4 ? (pitch0 >> 2) : pitch0;
    if (threadIdx.x == 0) {
      s_frame = f0; s_fp = fp0;
      s_magic[0] = slots0 > 1 ? 0xFFFFFFFFu / (uint32_t)slots0 + 1u : 0u;
      s_magic[1] = rslots0 > 1 ? 0xFFFFFFFFu / (uint32_t)rslots0 + 1u : 0u;
    }
  }
  __syncthreads();
  f = s_frame; fp = s_fp; m_slots = s_magic[0]; m_rslots = s_magic[1];
  }
  const long long spb64 = (long long)f.gh * f.gw;
  if (!lvl_ok || f.b < 0 || f.b >= B || spb64 <= 0) {
    // invalid batch / level index: zeros.  Empty sampling grid: the sum of nothing over the
    // count -- 0 (V2, rotated) or 0/0 = NaN (V1 divides by gh*gw, roi_align_kernel.cu:113)
    const float v = (!lvl_ok || f.b < 0 || f.b >= B) ? 0.0f : 0.0f / f.cnt;
    for (int it = threadIdx.x; it < (c_end - c_block) * nb; it += 256)
      out[(k * C + c_block) * (int64_t)nb + it] = v;
    return;
  }
  // LDS image of a footprint: row pitch = an ODD number of 16-byte slots (X4) / of floats.  With an even
  // number the rows of the image alias on the 32 banks (pitch 16 floats: two row phases; pitch 32: every
  // row on the same banks) and the y / y + 1 taps of neighbouring bins collide; the pad slot of a row is
  // loaded from the row's last chunk (a per-lane SOURCE address: the LDS side of LDS-DMA is linear).
  const int pitch = X4 ? (((fp.w >> 2) | 1) << 2) : (fp.w | 1);
  const int fsz = pitch * fp.h;                      // floats per channel plane in LDS
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : spb64);   // samples per bin
  const int nsamp = nb * spb;
  // block-uniform: huge RoI / huge sampling grid / >2^31-element tensors gather directly
  if (fsz > kFwdStageFloats || nsamp > kMaxTabSamples || (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_forward_direct(feat, C, H, W, f, k, ph, pw, c_block, c_end, out);
    return;
  }
  // ---- small footprint (the common case): every WAVE stages and reduces its own quarter of the block's
  // channels in a private LDS region -- no block barriers after the table's, so the four waves (and the other
  // blocks of the CU) drift apart and one wave's DMA latency hides behind another's gathers.  A lane owns one
  // bin: its <= 4 samples' tap offsets and weights are read from the table ONCE into registers and reused for
  // every channel.  The wave's FIRST pass is requested before the sample table is built: the copy depends on
  // the footprint only, and the table (trigonometry, ~100 instructions, a block barrier) is done while it travels.
  const bool small = fsz <= kWaveFloats && nb <= 64 && spb <= 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *wt = tile + wave * (kWaveFloats + kWavePad);
  const int per_wave = (c_end - c_block + 3) >> 2;
  const int cw0 = c_block + wave * per_wave, cw1 = min(c_end, cw0 + per_wave);
  const int cpw = small ? kWaveFloats / fsz : 1;     // channels per pass (>= 1)
  const uint32_t plane32w = (uint32_t)plane;
  // source offsets of a pass's DMA instructions: the same for every pass (only the channel base moves),
  // computed ONCE per wave -- slot -> (channel, row, chunk) costs two multiply-high divisions per lane
  // (a bound that depends on X4 -- `X4 ? 5 : 1` -- made hipcc 7.2 drop the kernel's host stub without a diagnostic)
  constexpr int kMaxInst = 5;                       // 16-byte path: <= kWaveFloats / 256 instructions per pass
  static_assert(kMaxInst * 256 == kWaveFloats, "one pass = at most kMaxInst 1-KiB DMA instructions");
  uint32_t doff[kMaxInst], dch[kMaxInst];
  const int slots = X4 ? (fsz >> 2) : fsz;           // 16-byte (dword) slots per channel
  const int rslots = X4 ? (pitch >> 2) : pitch;      // ... per row (the last one is the pad)
  const int rvalid = X4 ? (fp.w >> 2) : fp.w;        // ... that hold pixels
  auto slot_off = [&](uint32_t idx, uint32_t &c) -> uint32_t {
    c = slots > 1 ? __umulhi(idx, m_slots) : idx;
    const uint32_t r = idx - c * (uint32_t)slots;
    const uint32_t y = rslots > 1 ? __umulhi(r, m_rslots) : r;
    uint32_t x = r - y * (uint32_t)rslots;
    x = min(x, (uint32_t)(rvalid - 1));              // the pad slot re-loads the row's last chunk
    return y * (uint32_t)W + (X4 ? (x << 2) : x);
  };
  auto stage_pass = [&](int cp) {
    const int nch = min(cpw, cw1 - cp);
    const float *src = feat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    const int total = nch * slots;
    if (X4) {
#pragma unroll
      for (int j = 0; j < kMaxInst; ++j) {
        if (j * 64 < total) {                        // (wave-uniform)
          // a last, shorter pass: slots of channels it does not have re-load its last channel
          const uint32_t c = min(dch[j], (uint32_t)(nch - 1));
          __builtin_amdgcn_global_load_lds(src + (c * plane32w + doff[j]), (__attribute__((address_space(3))) void *)(wt + j * 256), 16, 0, 0);
        }
      }
    } else {
      for (int base = 0; base < total; base += 64) {
        uint32_t c;
        const uint32_t o = slot_off((uint32_t)min(base + lane, total - 1), c);
        __builtin_amdgcn_global_load_lds(src + (c * plane32w + o), (__attribute__((address_space(3))) void *)(wt + base), 4, 0, 0);
      }
    }
  };
  if (small) {
    if (X4) {
#pragma unroll
      for (int j = 0; j < kMaxInst; ++j) {
        const uint32_t idx = (uint32_t)min(j * 64 + lane, cpw * slots - 1);
        doff[j] = slot_off(idx, dch[j]);
      }
    }
    if (cw0 < cw1 && kAblate != 4) stage_pass(cw0);
  }
  OBB_STAMP(1);
  // sample table: coordinates, taps and weights are computed ONCE per RoI and block
  // (the reference recomputes them for every output element, roi_align_kernel.cu:70-113)
  for (int sidx = planned ? nsamp : (int)threadIdx.x; sidx < nsamp; sidx += 256) {     // (planned: the record carries the table)
    if constexpr (kAblate == 1) {   // (a plan record would carry the table)
      const unsigned short o = (unsigned short)((sidx * 7) % (fsz > 1 ? fsz - 1 : 1));
      s_off[sidx] = make_ushort4(o, o, (unsigned short)(o + 1 < fsz ? o + 1 : o), o);
      s_wgt[sidx] = make_float4(0.25f, 0.25f, 0.25f, 0.25f);
      continue;
    }
    const int bin = sidx / spb, q = sidx - bin * spb;
    const int bi = bin / pw, bj = bin - bi * pw;
    const int iy = q / f.gw, ix = q - iy * f.gw;
    float x, y, w[4];
    int yl, xl, yh, xh;
    sample_xy(f, bi, bj, iy, ix, x, y);
    const bool live = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
    // tap coordinates -> footprint-local offsets (clamped: never out of the tile)
    auto cl = [](int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); };
    const int ly0 = cl(yl - fp.y0, fp.h) * pitch, ly1 = cl(yh - fp.y0, fp.h) * pitch;
    const int lx0 = cl(xl - fp.x0, fp.w), lx1 = cl(xh - fp.x0, fp.w);
    s_off[sidx] = live ? make_ushort4((unsigned short)(ly0 + lx0), (unsigned short)(ly0 + lx1), (unsigned short)(ly1 + lx0), (unsigned short)(ly1 + lx1))
                       : make_ushort4(0, 0, 0, 0);
    s_wgt[sidx] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if (small) {
    int4 so[4];
    float4 sw[4];
    if (planned) {
      // every lane takes its bin's entries straight from the record ([sample][bin]: coalesced); the loads travel with
      // the first copy, both are waited for together below -- no LDS table, no block barrier
      const ushort4 *t_off = reinterpret_cast<const ushort4 *>(plan + (size_t)k * kPlanStride + kPlanHdrBytes);
      const float4 *t_wgt = reinterpret_cast<const float4 *>(plan + (size_t)k * kPlanStride + kPlanHdrBytes + kPlanTabEntries * sizeof(ushort4));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool on = lane < nb && q < spb;
        const ushort4 o16 = t_off[on ? q * 64 + lane : 0];
        const float4 w4 = t_wgt[on ? q * 64 + lane : 0];
        so[q] = on ? make_int4(o16.x, o16.y, o16.z, o16.w) : make_int4(0, 0, 0, 0);
        sw[q] = on ? w4 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      OBB_STAMP(2);
    } else {
    __syncthreads();   // table complete
    OBB_STAMP(2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool on = lane < nb && q < spb;
      const ushort4 o16 = s_off[on ? lane * spb + q : 0];
      so[q] = on ? make_int4(o16.x, o16.y, o16.z, o16.w) : make_int4(0, 0, 0, 0);
      sw[q] = on ? s_wgt[lane * spb + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    }
    // mean over the samples: a division, or -- a power-of-two count (2 x 2 samples), the usual case -- the
    // exact same value from a multiplication (scaling by a power of two rounds identically)
    const uint32_t cbits = __float_as_uint(f.cnt);
    const bool cnt_pow2 = (cbits & 0x007FFFFFu) == 0u && f.cnt >= 1.0f;
    const float cnt_inv = cnt_pow2 ? 1.0f / f.cnt : 0.0f;
    for (int cp = cw0; cp < cw1; cp += cpw) {
      const int nch = min(cpw, cw1 - cp);
      if (cp != cw0 && kAblate != 4) stage_pass(cp);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (cp == cw0) OBB_STAMP(3);
      __builtin_amdgcn_wave_barrier();
      auto bin_sum = [&](const float *p) -> float {
        if constexpr (kAblate == 3) return p[so[0].x] * cnt_inv;     // (one LDS read per output)
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
        return cnt_pow2 ? acc * cnt_inv : acc / f.cnt;
      };
      float *o = out + ((k * C + cp) * (int64_t)nb) + lane;
      // (two channels per step -- 2 x 16 LDS reads in flight together -- needs 116 VGPRs: 4 waves per SIMD instead of 5,
      // 41 -> 45 us; capped at 96 registers it spills: 79 us)
      for (int c = 0; c < nch; ++c) {
        const float r0 = bin_sum(wt + c * fsz);
        if constexpr (kAblate == 6) { if (lane < nb && r0 == 12345.678f) o[(int64_t)c * nb] = r0; }   // (no output stores)
        else if (lane < nb) o[(int64_t)c * nb] = r0;
      }
      // the next pass overwrites this region: its gathers are done (their values were consumed)
      __builtin_amdgcn_wave_barrier();
    }
    OBB_STAMP(4);
    return;
  }
  const int cpp = kFwdStageFloats / fsz;                // channels per pass (>= 1)
  const uint32_t m_nb = nb > 1 ? 0xFFFFFFFFu / (uint32_t)nb + 1u : 0u;  // it / nb for it < 2^16
  const uint32_t m_fsz = fsz > 1 ? 0xFFFFFFFFu / (uint32_t)fsz + 1u : 0u;
  const uint32_t m_w = pitch > 1 ? 0xFFFFFFFFu / (uint32_t)pitch + 1u : 0u;
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
        const int q4 = fsz >> 2, rw4 = pitch >> 2, total4 = total >> 2;   // slots per plane / row (incl. the pad) / pass
        const uint32_t m_q4 = q4 > 1 ? 0xFFFFFFFFu / (uint32_t)q4 + 1u : 0u;
        const uint32_t m_r4 = rw4 > 1 ? 0xFFFFFFFFu / (uint32_t)rw4 + 1u : 0u;
        const uint32_t xlast = (uint32_t)((fp.w >> 2) - 1);
        for (int base = wave * 64; base < total4; base += 256) {
          const uint32_t idx = (uint32_t)min(base + lane, total4 - 1);
          const uint32_t c = q4 > 1 ? __umulhi(idx, m_q4) : idx;
          const uint32_t r = idx - c * (uint32_t)q4;
          const uint32_t y = rw4 > 1 ? __umulhi(r, m_r4) : r;
          const uint32_t x = min(r - y * (uint32_t)rw4, xlast) << 2;   // (the pad slot re-loads the row's last chunk)
          const uint32_t off = c * plane32 + y * (uint32_t)W + x;
          __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void *)(tile + base * 4), 16, 0, 0);
        }
      } else {
        for (int base = wave * 64; base < total; base += 256) {
          const uint32_t idx = (uint32_t)min(base + lane, total - 1);  // tail lanes re-load the last float into the pad
          const uint32_t c = fsz > 1 ? __umulhi(idx, m_fsz) : idx;
          const uint32_t r = idx - c * (uint32_t)fsz;
          const uint32_t y = pitch > 1 ? __umulhi(r, m_w) : r;
          const uint32_t x = min(r - y * (uint32_t)pitch, (uint32_t)(fp.w - 1));
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
        const ushort4 l = s_off[bin * spb + q];
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
// The gradient of the map is a SCATTER of every output gradient to its 16 taps.  Scattering
// with LDS float atomics (the obvious mirror of the forward kernel) measured ~3 clocks per
// LANE on gfx950 -- 0.53 of the 0.59 ms this kernel used to take.  So the block turns the
// scatter into a GATHER once per RoI:
//   1. the sample table (as in the forward kernel) is inverted into a CSR list "pixel ->
//      (bin, weight / count) of every tap that lands on it" -- integer LDS atomics for the counts and
//      the slots, one block-wide scan; ~4 entries per pixel, 784 per RoI; done ONCE, it is
//      the same for every channel;
//   2. per pass over <= 32 channels: a lane is (pixel, channel), channel in the low lane
//      bits; it walks its pixel's list and sums g[channel][bin] * w / count out of LDS
//      (reads of consecutive channels = distinct banks), plain LDS store of the sum;
//   3. the touched pixels are added to the gradient map with global float atomics, in row
//      segments of consecutive addresses (RoIs overlap, so this part stays atomic).
// The order in which a pixel's contributions are summed follows the slot order (atomic
// tickets): results are reproducible to rounding only, like the reference's atomicAdd
// backward (roi_align_kernel_v2.cu:179-263); tolerance 1e-4 * max|g|.
constexpr int kBwdStageFloats = 4096;                  // sums tile (16 KiB): 16 channels x 255 pixels, 8 x 511, 4 x 1023

// plan record of the gather formulation (rroi_plan_kernel); state 1: planned, 2: left to the atomics kernel, 0: nothing sampled

__global__ __launch_bounds__(256) void rroi_backward_kernel(
    const float *__restrict__ gout, int B, int C, int H, int W, const float *__restrict__ rois,
    int ph, int pw, float scale, int sample_num, float min_size, int chan_per_block,
    int K, int chunks, int roi_kind, float *__restrict__ gfeat, int only_unstaged, const RoiPlan *__restrict__ plan) {
  __shared__ __attribute__((aligned(16))) float tile[kBwdStageFloats];   // first the sample table, then sums [channel][pixel]
  __shared__ int2 entry[kBwdMaxEntries];               // (bin, weight bits), grouped by pixel
  __shared__ int pstart[kBwdMaxPix + 1];               // CSR offsets; during the build: counts
  __shared__ float gtile[kMaxPassChan * kBinPitchMax];
  __shared__ int wsum[4];
  int *pfill = reinterpret_cast<int *>(tile) + kMaxTabSamples * 8;   // slot tickets of the build: behind the sample table
  static_assert(kBwdStageFloats >= kMaxTabSamples * 8 + kBwdMaxPix, "the sample table and the slot tickets fit the tile they are built in");
  int64_t k;
  int chunk;
  if (!block_to_work(K, chunks, k, chunk)) return;
  // (after the gather kernel: only what the plan left over.  One workgroup per item that leaves at this word is the
  // floor -- 4.3 us for 8192 items at configs[2]; 512 workgroups walking the items took 5.0 us, and a workgroup per 256
  // items that works off its flagged ones serially 4.2 us there but 0.48 instead of 0.36 ms at the config-4 shape,
  // whose 84 large RoIs x 16 chunks then share a few workgroups.)
  if (plan && plan[k].state != 2) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  const long long spb64 = (long long)f.gh * f.gw;
  if (f.b < 0 || f.b >= B || spb64 <= 0) return;     // nothing was sampled: no gradient
  const int nb = ph * pw;
  const int c_block = chunk * chan_per_block;
  const int c_end = min(C, c_block + chan_per_block);
  const int64_t plane = (int64_t)H * W;
  const Footprint fp = roi_footprint(f, H, W, ph, pw);
  const int fsz = fp.w * fp.h;
  const int cpitch = fsz | 1;
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : spb64);
  const int nsamp = nb * spb;
  if (fsz > kBwdMaxPix || cpitch > kBwdStageFloats || nb > kMaxBins || nsamp > kMaxTabSamples ||
      (int64_t)B * C * plane >= (1ll << 31)) {
    rroi_backward_direct(gout, C, H, W, f, k, ph, pw, c_block, c_end, gfeat);
    return;
  }
  if (only_unstaged) return;   // (the gather formulation below served this RoI)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- 1. sample table (in the tile's memory) and its inverse
  int4 *s_off = reinterpret_cast<int4 *>(tile);
  float4 *s_wgt = reinterpret_cast<float4 *>(tile) + kMaxTabSamples;
  for (int i = tid; i < fsz; i += 256) { pstart[i] = 0; pfill[i] = 0; }
  build_table(f, fp, H, W, pw, spb, nsamp, s_off, s_wgt);
  __syncthreads();
  const int ntap = nsamp * 4;
  auto tap_of = [&](int t, int &pix, float &w) {
    const int sidx = t >> 2, j = t & 3;
    const int4 o = s_off[sidx];
    const float4 ww = s_wgt[sidx];
    pix = j == 0 ? o.x : (j == 1 ? o.y : (j == 2 ? o.z : o.w));
    w = j == 0 ? ww.x : (j == 1 ? ww.y : (j == 2 ? ww.z : ww.w));
  };
  // (the taps of a bin that land on one pixel are merged into one entry, as in rroi_plan_kernel)
  const int tpb = 4 * spb;
  const bool merge = tpb <= 64 && (tpb & (tpb - 1)) == 0;
  constexpr int kTapIters = kBwdMaxEntries / 256;
  bool lead[kTapIters];
  float wsumq[kTapIters];
#pragma unroll
  for (int it = 0; it < kTapIters; ++it) {
    const int t = tid + it * 256;
    int pix = -1;
    float w = 0.0f;
    if (t < ntap) { tap_of(t, pix, w); w = w != 0.0f ? w / f.cnt : 0.0f; }   // weight / count: one division per tap and RoI
    if (w == 0.0f) pix = -1;                          // (a zero weight adds nothing, also in the direct kernel)
    lead[it] = pix >= 0;
    wsumq[it] = w;
    if (merge && tpb > 1 && __ballot(t < ntap)) merge_bin_taps(pix, w, lane, tpb, lead[it], wsumq[it]);
    if (lead[it]) atomicAdd(&pstart[pix], 1);
  }
  __syncthreads();
  {  // exclusive scan of the counts (<= 2048 pixels: 8 per thread)
    int c4[8], sum = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid * 8 + u; c4[u] = i < fsz ? pstart[i] : 0; sum += c4[u]; }
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid * 8 + u; if (i < fsz) pstart[i] = base; base += c4[u]; }
    if (tid == 255) pstart[fsz] = base;               // (tid 255 owns the last pixels: base = total)
    __syncthreads();
  }
#pragma unroll
  for (int it = 0; it < kTapIters; ++it) {
    const int t = tid + it * 256;
    if (lead[it]) {
      int pix;
      float w;
      tap_of(t, pix, w);
      const int slot = pstart[pix] + atomicAdd(&pfill[pix], 1);
      entry[slot] = make_int2(t / (4 * spb), __float_as_int(wsumq[it]));
    }
  }
  __syncthreads();   // entries complete; the table in `tile` is dead from here on
  // ---- 2. / 3. passes over the channels
  int cb = pass_channels(cpitch, c_end - c_block);
  while (cb > 1 && cpitch * cb > kBwdStageFloats) cb >>= 1;
  const int lc = lane & (cb - 1);
  const int psub = lane / cb, ppw = 64 / cb;         // pixel slot of this lane, pixels per wave step
  const int nbp = nb | 1;
  const uint32_t m_cp = 0xFFFFFFFFu / (uint32_t)cpitch + 1u;
  const uint32_t m_w = fp.w > 1 ? 0xFFFFFFFFu / (uint32_t)fp.w + 1u : 0u;
  const uint32_t m_nb = nb > 1 ? 0xFFFFFFFFu / (uint32_t)nb + 1u : 0u;
  for (int cp = c_block; cp < c_end; cp += cb) {
    const int nch = min(cb, c_end - cp);
    const float *gsrc = gout + (k * C + cp) * (int64_t)nb;   // contiguous (channel, bin) run
    const int gtot = nch * nb;
    for (int base = 0; base < gtot; base += 256 * 4) {        // (4 loads in flight per thread)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = gsrc[min(base + u * 256 + tid, gtot - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + tid;
        if (i < gtot) {
          const int c = nb > 1 ? (int)__umulhi((uint32_t)i, m_nb) : i, bin = i - c * nb;
          gtile[c * nbp + bin] = v[u];
        }
      }
    }
    __syncthreads();
    const float *gl = gtile + lc * nbp;
    for (int p0 = wave * ppw; p0 < fsz; p0 += 4 * ppw) {
      const int pix = p0 + psub;
      if (pix < fsz && lc < nch) {
        const int e0 = pstart[pix], e1 = pstart[pix + 1];
        float acc = 0.0f;
        for (int e = e0; e < e1; ++e) {
          const int2 en = entry[e];
          acc = acc + gl[en.x] * __int_as_float(en.y);   // g * (w / count)
        }
        tile[lc * cpitch + pix] = acc;
      }
    }
    __syncthreads();
    float *dst = gfeat + ((int64_t)f.b * C + cp) * plane + (int64_t)fp.y0 * W + fp.x0;
    const int total = nch * cpitch;
    for (int idx = tid; idx < total; idx += 256) {
      const uint32_t c = cpitch > 1 ? __umulhi((uint32_t)idx, m_cp) : (uint32_t)idx;
      const uint32_t p = (uint32_t)idx - c * (uint32_t)cpitch;
      if (p < (uint32_t)fsz) {                        // (not the pitch's pad word)
        const float v = tile[idx];
        if (v != 0.0f) {
          const uint32_t y = fp.w > 1 ? __umulhi(p, m_w) : p;
          const uint32_t x = p - y * (uint32_t)fp.w;
          atomicAdd(dst + (int64_t)c * plane + (int64_t)y * W + x, v);
        }
      }
    }
    __syncthreads();   // the next pass rewrites gtile and the tile
  }
}

// ---- backward, GATHER formulation (needs a workspace; deterministic) ------------------------------
// The scatter above ends in ~1.6e6 atomic row requests per launch (one per channel and footprint row),
// the floor of that design.  With a workspace the whole backward becomes a gather from the MAP side:
//   plan    (one block per RoI)   frame, footprint and the inverted sample table "pixel -> (bin,
//           weight / count) of every tap on it" -- the prologue of the kernel above, built ONCE per RoI
//           instead of once per channel chunk, the taps of a pixel sorted by tap number;
//   gather  (one block per 16 x 16 map tile and 16 channels)  the RoIs whose footprint meets the tile,
//           found by testing all K plans, hence in RoI order; every thread owns one pixel and walks
//           the lists of that pixel in all those RoIs, 16 channel sums in registers; ONE plain store
//           per map element, which is written whether a RoI touched it or not (no zero fill needed).
// No atomics, and a fixed summation order (RoI index, then tap number): results are reproducible bit
// for bit, which the reference's atomicAdd backward (roi_align_kernel_v2.cu:179-263) is not.
// RoIs the plan cannot hold (footprint > 2047 pixels, > 256 samples, > 64 bins) are added afterwards by
// the atomics kernel (`only_unstaged`).
constexpr int kGTile = 16;                          // map tile: 16 x 16 pixels = one pixel per thread
constexpr int kGChanDefault = 16;                   // channels per gather block (sums in registers); OBB_TUNE_RROI_GATHER_CHAN
constexpr int kGList = 2048;                        // RoIs of one tile held in LDS at a time

// OBB_PLAN_STAMPS (tools/build_ablations.sh -> libobbops_plst.so, tools/exp_plan_stamps.py): shader-clock stamps of the
// plan kernel's phases, one row of 12 per block (10 / 11: the 100 MHz real-time counter at entry / exit)
#ifdef OBB_PLAN_STAMPS
__device__ unsigned long long g_plan_stamps[4096 * 12];
#define OBB_PSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_plan_stamps[blockIdx.x * 12 + (i)] = (i) >= 10 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define OBB_PSTAMP(i) do { } while (0)
#endif

// plan blocks: 4 waves.  (8 waves measured in round 5, tools/exp_plan_stamps.py: the tap phase 7250 -> 4450 cycles, but frame /
// table / finish each ~500 slower with two waves per SIMD: block life 20.6k -> 18.0k cycles, whole backward 56.7 -> 57.5 us.)
constexpr int kPlanThreads = 256, kPlanWaves = kPlanThreads / 64;
template <bool UNUSED = false>
__device__ __forceinline__ bool plan_holds(int fsz, int nb, int nsamp) {
  return fsz <= kBwdMaxPix && (fsz | 1) <= kBwdStageFloats && nb <= kMaxBins && nsamp <= kMaxTabSamples;
}

__global__ __launch_bounds__(kPlanThreads) void rroi_plan_kernel(
    int B, int H, int W, const float *__restrict__ rois, int ph, int pw, float scale, int sample_num,
    float min_size, int roi_kind, RoiPlan *__restrict__ plan, unsigned short *__restrict__ rstart,
    int2 *__restrict__ rentry, int *__restrict__ lpt, int tiles_x, int tiles_per_image,
    int nplan, const float *__restrict__ tg, float *__restrict__ tgT, int tC) {
  __shared__ __attribute__((aligned(16))) float tile[kBwdStageFloats];   // sample table | per-pixel counts
  if ((int)blockIdx.x >= nplan) {
    // (channels-last gather, obb_rroi_nhwc.hip) blocks behind the plan's transpose grad_out (K, C, nb) -> tgT (K, nb, C),
    // one (RoI, 64 channels) each, through `tile` ([64][<= 63 bins], odd pitch): independent of the plan blocks, whose
    // dependent chains leave the memory pipes idle beside them
    const int nb = ph * pw, chunks = (tC + 63) / 64;
    const int id = (int)blockIdx.x - nplan, k = id / chunks, c0 = (id - k * chunks) * 64, cc = min(64, tC - c0);
    const float *src = tg + ((int64_t)k * tC + c0) * nb;
    float *dst = tgT + (int64_t)k * nb * tC + c0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kTB = 63;                                   // bins per pass: 64 x 63 floats fit the tile
    if (nb <= kTB && !(tC & 3) && !(reinterpret_cast<uintptr_t>(tg) & 15u)) {
      // the block's (channel, bin) slab is ONE contiguous, 16-byte aligned run of cc * nb floats: every request issued
      // before the first is waited for (the loop below waits for each load before its LDS store -- 13 round trips in
      // a row made these blocks, not the plan's, the end of the launch: 19.0 us)
      const int n4 = (cc * nb) >> 2;                          // <= 1008 float4
      const float4 *src4 = reinterpret_cast<const float4 *>(src);
      constexpr int kV = 1024 / kPlanThreads;
      float4 v[kV];
#pragma unroll
      for (int u = 0; u < kV; ++u) {
        const int i4 = (int)threadIdx.x + u * kPlanThreads;
        v[u] = i4 < n4 ? src4[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < kV; ++u) {
        const int i4 = (int)threadIdx.x + u * kPlanThreads;
        if (i4 < n4) {
          int c = (i4 * 4) / nb, bb = i4 * 4 - c * nb;
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            tile[c * kTB + bb] = e[q];
            if (++bb == nb) { bb = 0; ++c; }
          }
        }
      }
      __syncthreads();
      // a quarter wave per bin row: lane -> 4 channels (cc is a multiple of 4), one 16-byte store each
      const int l16 = lane & 15, sub = lane >> 4;
      for (int bb = wave * 4 + sub; bb < nb; bb += 4 * kPlanWaves)
        if (l16 * 4 < cc) {
          const float *t = tile + (l16 * 4) * kTB + bb;
          *reinterpret_cast<float4 *>(dst + (int64_t)bb * tC + l16 * 4) = make_float4(t[0], t[kTB], t[2 * kTB], t[3 * kTB]);
        }
      return;
    }
    for (int b0 = 0; b0 < nb; b0 += kTB) {
      const int bn = min(kTB, nb - b0);
      for (int i = threadIdx.x; i < cc * bn; i += kPlanThreads) {
        const int c = i / bn, bb = i - c * bn;
        tile[c * kTB + bb] = src[(int64_t)c * nb + b0 + bb];
      }
      __syncthreads();
      for (int bb = wave; bb < bn; bb += kPlanWaves)
        if (lane < cc) dst[(int64_t)(b0 + bb) * tC + lane] = tile[lane * kTB + bb];
      __syncthreads();
    }
    return;
  }
  __shared__ int s_last;
  __shared__ int2 entry[kBwdMaxEntries];               // (bin | tap << 8, weight bits), grouped by pixel
  __shared__ int pstart[kBwdMaxPix + 1];
  __shared__ int wsum[kPlanWaves];
  int *pfill = reinterpret_cast<int *>(tile) + kMaxTabSamples * 8;
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  OBB_PSTAMP(10);
  OBB_PSTAMP(0);
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  const long long spb64 = (long long)f.gh * f.gw;
  RoiPlan pl = {f.b, 0, 0, 0, 0, 0, ph * pw, 0};
  const bool live = f.b >= 0 && f.b < B && spb64 > 0;
  const int nb = ph * pw;
  const Footprint fp = live ? roi_footprint(f, H, W, ph, pw) : Footprint{0, 0, 0, 0};
  const int fsz = fp.w * fp.h;
  const int spb = (int)(spb64 > kMaxTabSamples ? kMaxTabSamples + 1 : (spb64 > 0 ? spb64 : 1));
  const int nsamp = nb * spb;
  // Heavy map tiles first.  The gather kernel's blocks take one (map tile, channel chunk) each and last as long as
  // RoIs meet the tile (2 .. 16 at configs[2]); in tile order the launch ends when the last heavy tile, started late,
  // is done.  Every planned RoI counts itself into the tiles its footprint meets; the LAST plan block to finish (a
  // ticket) orders every image's tiles by descending count (a counting sort) for the gather kernel's block -> tile map.
  auto finish = [&]() {
    if (!lpt) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this wave's counts are acknowledged before the ticket is taken)
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(lpt + kLptMax, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nplan - 1;
    __syncthreads();
    if (!s_last) return;
    int *bucket = pstart;                    // [256] tiles per count (255 = "255 or more"), then their first rank
    int *perm = lpt + kLptMax + 16;
    for (int img = 0; img < B; ++img) {
      __syncthreads();
      if (tid < 256) bucket[tid] = 0;
      __syncthreads();
      const int *cnt = lpt + img * tiles_per_image;
      for (int t = tid; t < tiles_per_image; t += kPlanThreads) {
        const int c = __hip_atomic_load(cnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&bucket[c > 255 ? 255 : c], 1);
      }
      __syncthreads();
      {  // first rank of every count, heaviest first: exclusive scan over the buckets in descending order
        const int mine = tid < 256 ? bucket[255 - tid] : 0;   // (256 buckets: the first four waves)
        int inc = mine;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = inc - mine;
        for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
        if (tid < 256) bucket[255 - tid] = base;
      }
      __syncthreads();
      for (int t = tid; t < tiles_per_image; t += kPlanThreads) {
        const int c = __hip_atomic_load(cnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        perm[img * tiles_per_image + atomicAdd(&bucket[c > 255 ? 255 : c], 1)] = t;
      }
    }
  };
  if (!live || !plan_holds(fsz, nb, nsamp)) {        // nothing sampled, or left to the atomics kernel
    pl.state = live ? 2 : 0;
    if (tid == 0) plan[k] = pl;
    finish();
    return;
  }
  if (lpt) {
    const int tx0 = fp.x0 / kGTile, tx1 = (fp.x0 + fp.w - 1) / kGTile, ty0 = fp.y0 / kGTile, ty1 = (fp.y0 + fp.h - 1) / kGTile;
    const int nx = tx1 - tx0 + 1, nt = nx * (ty1 - ty0 + 1);
    for (int i = tid; i < nt; i += kPlanThreads) atomicAdd(&lpt[f.b * tiles_per_image + (ty0 + i / nx) * tiles_x + tx0 + i % nx], 1);
  }
  int4 *s_off = reinterpret_cast<int4 *>(tile);
  float4 *s_wgt = reinterpret_cast<float4 *>(tile) + kMaxTabSamples;
  OBB_PSTAMP(1);
  for (int i = tid; i < fsz; i += kPlanThreads) { pstart[i] = 0; pfill[i] = 0; }
  build_table<kPlanThreads>(f, fp, H, W, pw, spb, nsamp, s_off, s_wgt);
  __syncthreads();
  OBB_PSTAMP(2);
  const int ntap = nsamp * 4;
  auto tap_of = [&](int t, int &pix, float &w) {
    const int sidx = t >> 2, j = t & 3;
    const int4 o = s_off[sidx];
    const float4 ww = s_wgt[sidx];
    pix = j == 0 ? o.x : (j == 1 ? o.y : (j == 2 ? o.z : o.w));
    w = j == 0 ? ww.x : (j == 1 ? ww.y : (j == 2 ? ww.z : ww.w));
  };
  // The taps of one bin that land on the SAME pixel become one entry (the sum of their weights / count, in tap order):
  // a small RoI -- bins of a fraction of a pixel -- puts 3-4 of a bin's 16 taps on each of ~5 pixels, and the gather
  // kernel pays per entry (512 RoIs of 8-16 px: 140 us of gather against 50 us for 32-128 px).  A bin's taps are
  // consecutive tap numbers, i.e. consecutive lanes of one iteration here (when 4 x samples-per-bin is a power of two
  // <= 64: groups of that many lanes); the first tap of a (bin, pixel) group is its leader and carries the sum.
  const int tpb = 4 * spb;                                   // taps per bin
  const bool merge = tpb <= 64 && (tpb & (tpb - 1)) == 0;
  constexpr int kTapIters = kBwdMaxEntries / kPlanThreads;   // <= 4 taps per thread
  bool lead[kTapIters];
  float wsumq[kTapIters];
  int pixq[kTapIters];
#pragma unroll
  for (int it = 0; it < kTapIters; ++it) {
    const int t = tid + it * kPlanThreads;
    int pix = -1;
    float w = 0.0f;
    if (t < ntap) { tap_of(t, pix, w); w = w != 0.0f ? w / f.cnt : 0.0f; }
    if (w == 0.0f) pix = -1;                                 // (a zero weight adds nothing, also in the direct kernel)
    pixq[it] = pix;
    lead[it] = pix >= 0;
    wsumq[it] = w;
  }
  if (merge && tpb == 16) {
    // the usual 2 x 2 samples: the four rounds' merges back to back with no branch between them -- four independent chains
    // of 15 DPP steps that the scheduler interleaves (with a ballot and a branch per round they ran one after the other:
    // 7.3 k of the block's 20 k cycles); a round without taps holds pix = -1 everywhere and merges nothing
#pragma unroll
    for (int it = 0; it < kTapIters; ++it) merge_bin_taps(pixq[it], wsumq[it], lane, 16, lead[it], wsumq[it]);
  } else if (merge && tpb > 1) {
#pragma unroll
    for (int it = 0; it < kTapIters; ++it)
      if (__ballot(tid + it * kPlanThreads < ntap)) merge_bin_taps(pixq[it], wsumq[it], lane, tpb, lead[it], wsumq[it]);   // (wave-uniform)
  }
#pragma unroll
  for (int it = 0; it < kTapIters; ++it)
    if (lead[it]) atomicAdd(&pstart[pixq[it]], 1);
  __syncthreads();
  OBB_PSTAMP(3);
  {  // exclusive scan of the counts (<= 2048 pixels: 8 per thread)
    constexpr int kPer = 2048 / kPlanThreads;
    int c8[kPer], sum = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) { const int i = tid * kPer + u; c8[u] = i < fsz ? pstart[i] : 0; sum += c8[u]; }
    int inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
#pragma unroll
    for (int u = 0; u < kPer; ++u) { const int i = tid * kPer + u; if (i < fsz) pstart[i] = base; base += c8[u]; }
    if (tid == kPlanThreads - 1) pstart[fsz] = base;
    __syncthreads();
  }
  OBB_PSTAMP(4);
#pragma unroll
  for (int it = 0; it < kTapIters; ++it) {
    const int t = tid + it * kPlanThreads;
    if (lead[it]) {
      int pix;
      float w;
      tap_of(t, pix, w);
      const int slot = pstart[pix] + atomicAdd(&pfill[pix], 1);
      entry[slot] = make_int2((t / (4 * spb)) | (t << 8), __float_as_int(wsumq[it]));
    }
  }
  __syncthreads();
  OBB_PSTAMP(5);
  // the slots of a pixel were handed out in arrival order; the plan stores every pixel's taps in TAP order
  // so that the sums of the gather kernel do not depend on it.  A wave takes a pixel at a time, a lane one
  // of its entries and counts the entries with a smaller tap number (the lists are short: ~4 entries on
  // average; an insertion sort per thread took 0.25 ms on a 2 x 2 pixel RoI with ~90 taps per pixel)
  unsigned short *rs = rstart + k * (int64_t)kPlanPix;
  int2 *re = rentry + k * (int64_t)kBwdMaxEntries;
  for (int i = tid; i <= fsz; i += kPlanThreads) rs[i] = (unsigned short)pstart[i];
  // (a) a thread per pixel for the usual short lists (<= 8 taps: ranks by compares in registers) -- a wave per
  //     pixel walked the 2047 pixels of a large footprint one LDS round trip at a time: 0.1 ms for one RoI;
  // (b) the few longer lists (small RoIs: many taps per pixel) are queued and taken by whole waves.
  __shared__ int longq[kBwdMaxEntries / 8 + 1];
  __shared__ int nlong;
  if (tid == 0) nlong = 0;
  __syncthreads();
  for (int pix = tid; pix < fsz; pix += kPlanThreads) {
    const int e0 = pstart[pix], m = pstart[pix + 1] - e0;
    if (m > 8) { longq[atomicAdd(&nlong, 1)] = pix; continue; }
    int2 en[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) en[u] = u < m ? entry[e0 + u] : make_int2(0x7FFFFFFF, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < m) {
        int rank = 0;
#pragma unroll
        for (int v = 0; v < 8; ++v) rank += ((en[v].x >> 8) < (en[u].x >> 8)) ? 1 : 0;
        re[e0 + rank] = make_int2(en[u].x & 0xFF, en[u].y);
      }
    }
  }
  __syncthreads();
  OBB_PSTAMP(6);
  unsigned int *bm = reinterpret_cast<unsigned int *>(tile) + wave * 64;   // (the sample table is dead: 32 bitmap words + 32 prefixes per wave)
  const int nq = nlong;
  for (int qi = __builtin_amdgcn_readfirstlane(wave); qi < nq; qi += kPlanWaves) {
    const int pix = __builtin_amdgcn_readfirstlane(longq[qi]);
    const int e0 = __builtin_amdgcn_readfirstlane(pstart[pix]);   // (wave-uniform: tell the compiler, or the
    const int m = __builtin_amdgcn_readfirstlane(pstart[pix + 1]) - e0;   //  v_readlane below becomes a waterfall loop)
    if (m <= 64) {
      // one entry per lane: the rank is counted over the lanes' registers (no memory in the loop)
      int2 mine = make_int2(0x7FFFFFFF, 0);
      if (lane < m) mine = entry[e0 + lane];
      const int key = mine.x >> 8;
      int rank = 0;
      for (int j = 0; j < m; ++j) rank += (__builtin_amdgcn_readlane(key, j) < key) ? 1 : 0;
      if (lane < m) re[e0 + rank] = make_int2(mine.x & 0xFF, mine.y);
    } else {
      // a pixel of a tiny RoI holds up to all 1024 taps: rank = number of set bits below the tap's own
      // in a 1024-bit map of the taps present (cost independent of m)
      if (lane < 32) bm[lane] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int l = lane; l < m; l += 64) {
        const int key = entry[e0 + l].x >> 8;
        atomicOr(&bm[key >> 5], 1u << (key & 31));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      {
        const unsigned int wv = lane < 32 ? bm[lane] : 0u;
        int inc = __popc(wv);
        const int own = inc;
        for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane < 32) bm[32 + lane] = (unsigned int)(inc - own);     // set bits in the words before this one
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int l = lane; l < m; l += 64) {
        const int2 mine = entry[e0 + l];
        const int key = mine.x >> 8;
        const int rank = (int)bm[32 + (key >> 5)] + __popc(bm[key >> 5] & ((1u << (key & 31)) - 1u));
        re[e0 + rank] = make_int2(mine.x & 0xFF, mine.y);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  OBB_PSTAMP(7);
  if (tid == 0) { pl.x0 = fp.x0; pl.y0 = fp.y0; pl.w = fp.w; pl.h = fp.h; pl.state = 1; plan[k] = pl; }
  finish();
  OBB_PSTAMP(8);
  OBB_PSTAMP(11);
}

// Every global request of a RoI is issued one RoI ahead: while RoI i is summed, the plan record of RoI i+2, the
// list bounds and the gradients of RoI i+1 (in registers) are in flight, and the first four entries of RoI i+1 are
// requested as soon as its bounds are there -- a RoI costs one barrier and its sums, no exposed round trip.
// Every one of these loads is UNCONDITIONAL (clamped to an address that is always valid, the predicate applied where
// the value is used): a load under a branch is waited for at the end of the branch, which serialises the requests
// (round 2's kernel had four such round trips per RoI: 156 us at configs[2]; this one 98 us -- the rest of the step
// came from the compile-time bin count, the scalar RoI index and fused multiply-adds: 2434 -> ~1000 VALU per wave).
// (Also measured and dropped in round 3: FOUR RoIs per step, two barriers per step: 155 -> 185 us, 99 VGPRs.)
// (Small RoIs are this kernel's slow case -- tools/exp_plan.py: 512 RoIs of 8-16 px 140 us, of 32-128 px 50 us: a 3 x 3
// pixel footprint holds all 784 taps, ~90 per pixel, summed by 9 lanes of a wave.  Measured and dropped: the whole wave
// summing a long list (lane = (entry slot, channel), two shuffles per entry: as many instructions as it saves), rounds
// of 8 entries with the next round in flight (88 VGPRs, no gain: the lists are instruction-, not latency-bound).  What
// would help is merging a pixel's consecutive taps of one bin in the plan: ~5x fewer entries for such RoIs.)
// (Also measured and dropped: per-tile RoI lists appended by the plan blocks and sorted here, instead of every block testing
// all K plans: 0.0678 -> 0.0678 ms at configs[2], 0.319 -> 0.316 at the config-4 shape.)
// Grid = (chunks, tiles, images): the channel chunks of a map tile run together (they read the same plan records, list
// bounds and entries: 0.117 -> 0.108 ms at configs[2], 0.42 -> 0.385 at the config-4 shape against tiles fastest), the
// tiles heavy first (rroi_plan_kernel's order: -> 0.107 / 0.370 ms).
// NB: bins per RoI when known at compile time (49: the LDS reads of the sums take immediate offsets), 0: from the plan.
template <int CH, int NB>
__global__ __launch_bounds__(256) void rroi_gather_kernel(
    const float *__restrict__ gout, int C, int H, int W, int K, const RoiPlan *__restrict__ plan,
    const unsigned short *__restrict__ rstart, const int2 *__restrict__ rentry, float *__restrict__ gfeat,
    const int *__restrict__ tile_order) {
  constexpr int G = CH * kMaxBins / 256;            // gradient words a thread stages per RoI
  static_assert(G * 256 == CH * kMaxBins, "gradient staging");
  __shared__ int list[kGList];
  __shared__ float gtiles[2][CH * (kMaxBins + 1)];
  __shared__ int wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (W + kGTile - 1) / kGTile;
  const int b = (int)blockIdx.z;
  // grid (chunks, tiles, images): the channel chunks of a tile start together, the tiles in rroi_plan_kernel's order
  // (heavy first; any order gives the same sums)
  const int tl = tile_order ? tile_order[b * (int)gridDim.y + (int)blockIdx.y] : (int)blockIdx.y;
  const int tx0 = (tl % tiles_x) * kGTile, ty0 = (tl / tiles_x) * kGTile;
  const int c0 = (int)blockIdx.x * CH, cc = min(CH, C - c0);
  const int x = tx0 + (tid & (kGTile - 1)), y = ty0 + tid / kGTile;
  const bool inmap = x < W && y < H;
  float acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0.0f;
  struct Foot { int x0, y0, w, h, nb; };
  auto foot = [&](int r) { const RoiPlan p = plan[r]; return Foot{p.x0, p.y0, p.w, p.h, NB ? NB : p.nb}; };   // (r is wave-uniform: scalar loads)
  auto pick = [&](int i) { return __builtin_amdgcn_readfirstlane(list[i]); };
  // list bounds of this thread's pixel (pixel 0 of the footprint when the pixel is outside: masked by `mine` later)
  auto mine_of = [&](const Foot &f) {
    const int px = x - f.x0, py = y - f.y0;
    return inmap && px >= 0 && px < f.w && py >= 0 && py < f.h;
  };
  auto bounds = [&](int r, const Foot &f, bool mine, int &e0, int &e1) {
    const int pix = mine ? (y - f.y0) * f.w + (x - f.x0) : 0;
    const unsigned short *rs = rstart + (int64_t)r * kPlanPix + pix;
    e0 = rs[0]; e1 = rs[1];
  };
  for (int base = 0; base < K; base += kGList) {
    int n = 0;   // block-uniform
    const int lim = min(K, base + kGList);
    RoiPlan pq = plan[min(base + tid, lim - 1)];
    for (int r0 = base; r0 < lim; r0 += 256) {
      const int r = r0 + tid;
      const RoiPlan p = pq;
      pq = plan[min(r + 256, lim - 1)];          // (the next round's records travel during this round's barriers)
      const bool hit = r < lim && p.state == 1 && p.b == b && p.x0 < tx0 + kGTile && p.x0 + p.w > tx0 && p.y0 < ty0 + kGTile && p.y0 + p.h > ty0;
      const uint64_t mk = __ballot(hit);
      if (lane == 0) wcnt[wave] = __popcll(mk);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) { const int v = wcnt[w2]; off += (w2 < wave) ? v : 0; tot += v; }
      if (hit) list[n + off + __popcll(mk & ((1ull << lane) - 1ull))] = r;
      n += tot;
      __syncthreads();
    }
    if (n == 0) continue;
    // ---- prologue: everything of RoI 0, the plan record of RoI 1
    int r_cur = pick(0), r_nxt = pick(min(1, n - 1));
    Foot f_cur = foot(r_cur), f_nxt = foot(r_nxt);
    bool m_cur = mine_of(f_cur);
    int e0_cur, e1_cur;
    bounds(r_cur, f_cur, m_cur, e0_cur, e1_cur);
    float g_cur[G];
    {
      const float *gsrc = gout + ((int64_t)r_cur * C + c0) * f_cur.nb;
      const int last = cc * f_cur.nb - 1;
#pragma unroll
      for (int u = 0; u < G; ++u) g_cur[u] = gsrc[min(tid + 256 * u, last)];
    }
    int2 en_cur[4];
    {
      const int2 *re = rentry + (int64_t)r_cur * kBwdMaxEntries;
      const int first = m_cur ? e0_cur : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) en_cur[u] = re[min(first + u, kBwdMaxEntries - 1)];
    }
    for (int i = 0; i < n; ++i) {
      float *gtile = gtiles[i & 1];
      const int nb = f_cur.nb, nbp = nb | 1;
      // stage this RoI's gradients (requested one RoI ago)
      {
        const int gtot = cc * nb;
        if (nb & 1) {            // pitch == bins: a straight copy
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int q = tid + 256 * u;
            if (q < gtot) gtile[q] = g_cur[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int q = tid + 256 * u;
            if (q < gtot) { const int c = q / nb, bin = q - c * nb; gtile[c * nbp + bin] = g_cur[u]; }
          }
        }
      }
      // requests of the next RoI (bounds + gradients) and of the one after (plan record); past the end of the
      // list they repeat the last RoI's and are never used
      const bool m_nxt = mine_of(f_nxt);
      int e0_nxt, e1_nxt;
      bounds(r_nxt, f_nxt, m_nxt, e0_nxt, e1_nxt);
      {
        const float *gsrc = gout + ((int64_t)r_nxt * C + c0) * f_nxt.nb;
        const int last = cc * f_nxt.nb - 1;
#pragma unroll
        for (int u = 0; u < G; ++u) g_cur[u] = gsrc[min(tid + 256 * u, last)];
      }
      const int r_nn = pick(min(i + 2, n - 1));
      const Foot f_nn = foot(r_nn);
      __syncthreads();
      // sums of this RoI: the first four entries are already here
      const int e0 = m_cur ? e0_cur : 0, e1 = m_cur ? e1_cur : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (e0 + u < e1) {
          const float wgt = __int_as_float(en_cur[u].y);
          const float *g = gtile + en_cur[u].x;
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[c] = fmaf(g[c * nbp], wgt, acc[c]);
        }
      }
      if (e0 + 4 < e1) {
        const int2 *re = rentry + (int64_t)r_cur * kBwdMaxEntries;
        for (int e = e0 + 4; e < e1; e += 4) {
          int2 en[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) en[u] = re[min(e + u, e1 - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e + u < e1) {
              const float wgt = __int_as_float(en[u].y);
              const float *g = gtile + en[u].x;
#pragma unroll
              for (int c = 0; c < CH; ++c) acc[c] = fmaf(g[c * nbp], wgt, acc[c]);
            }
          }
        }
      }
      // the next RoI's first entries (its bounds were requested before the barrier)
      {
        const int2 *rn = rentry + (int64_t)r_nxt * kBwdMaxEntries;
        const int first = m_nxt ? e0_nxt : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) en_cur[u] = rn[min(first + u, kBwdMaxEntries - 1)];
      }
      r_cur = r_nxt; f_cur = f_nxt; m_cur = m_nxt; e0_cur = e0_nxt; e1_cur = e1_nxt;
      r_nxt = r_nn; f_nxt = f_nn;
    }
    __syncthreads();   // (the next batch of plans rewrites the list)
  }
  if (inmap) {
    float *dst = gfeat + (((int64_t)b * C + c0) * H + y) * (int64_t)W + x;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < cc) __builtin_nontemporal_store(acc[c], dst + (int64_t)c * H * W);   // (streams past the L2: the gradients and the plan stay in it)
  }
}

// ---- order of the RoIs (forward, with a workspace) ------------------------------------------------------------
// key = size class (largest first) | pyramid level | Morton code of the RoI centre on a 32 x 32 grid over its map;
// ONE workgroup counts, scans and scatters (32 768 bins in LDS): order[rank] = RoI.  RoIs that run at the same time on
// an XCD then share footprint lines in its L2 (Morton), and the RoIs whose blocks last longest -- the classes follow
// the number of staging passes a footprint needs -- start first instead of last (tools/exp_rroi_order2.py: config-4
// shape 220 us in the given order, 177 Morton only, 169 with the size classes).  The order inside a cell follows the
// LDS atomics' arrival -- it decides which block works on which RoI, never a result.
constexpr int kOrderBins = 32768;    // 4 size classes x 8 levels x 1024 cells
// below: not worth its launch (measured: 512 RoIs covering a 256^2 map 1.8x: the ordered kernel is 3 us faster, the order
// kernel and its launch boundary cost 8; 2000 RoIs covering it 22x: 220 -> 175 us all in)
constexpr int kOrderMinK = 1024;

__device__ __forceinline__ uint32_t roi_order_key(const float *__restrict__ roi, int kind, int lvl, const LevelTable &lv) {
  const float cx = kind == 0 ? roi[1] : 0.5f * (roi[1] + roi[3]);
  const float cy = kind == 0 ? roi[2] : 0.5f * (roi[2] + roi[4]);
  const float sc = lv.scale[lvl];
  float ux = cx * sc * 32.0f / (float)max(lv.W[lvl], 1), uy = cy * sc * 32.0f / (float)max(lv.H[lvl], 1);
  ux = ux >= 0.0f ? (ux < 31.0f ? ux : 31.0f) : 0.0f;      // (NaN -> 0)
  uy = uy >= 0.0f ? (uy < 31.0f ? uy : 31.0f) : 0.0f;
  uint32_t x = (uint32_t)ux, y = (uint32_t)uy;
  x = (x | (x << 4)) & 0x0F0Fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u;
  y = (y | (y << 4)) & 0x0F0Fu; y = (y | (y << 2)) & 0x3333u; y = (y | (y << 1)) & 0x5555u;
  // footprint estimate in map pixels -> class 0 (> 640: one channel per pass) .. 3 (<= 160: eight channels per pass)
  const float w = kind == 0 ? roi[3] : roi[3] - roi[1] + 1.0f, h = kind == 0 ? roi[4] : roi[4] - roi[2] + 1.0f;
  const float fpx = (w * sc + 2.0f) * (h * sc + 2.0f);
  const uint32_t cls = fpx > 640.0f ? 0u : (fpx > 320.0f ? 1u : (fpx > 160.0f ? 2u : 3u));   // (NaN -> 3)
  return (cls << 13) | ((uint32_t)lvl << 10) | x | (y << 1);
}

__global__ __launch_bounds__(1024) void rroi_order_kernel(const LevelTable lv, const int *__restrict__ roi_level,
                                                          int num_levels, const float *__restrict__ rois, int K,
                                                          int roi_kind, int *__restrict__ order, int *__restrict__ tmp) {
  __shared__ uint32_t hist[kOrderBins + kOrderBins / 32];   // bin i at i + i / 32: a thread's 32 consecutive bins on 32 banks
  auto P = [](uint32_t i) { return i + (i >> 5); };
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kOrderBins + kOrderBins / 32; i += 1024) hist[i] = 0u;
  __syncthreads();
  const int S = roi_stride(roi_kind);
  auto key_of = [&](int k) {
    const int l = roi_level ? roi_level[k] : 0;
    return roi_order_key(rois + (int64_t)k * S, roi_kind, (l >= 0 && l < num_levels) ? (l & 7) : 0, lv);
  };
  // (key, rank inside the cell) of a thread's first four RoIs stay in registers; only RoIs beyond 4096 go through `tmp`
  uint32_t rkey[4], rrank[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = tid + u * 1024;
    rkey[u] = 0u; rrank[u] = 0u;
    if (k < K) { rkey[u] = key_of(k); rrank[u] = atomicAdd(&hist[P(rkey[u])], 1u); }
  }
  for (int k = tid + 4096; k < K; k += 1024) tmp[k] = (int)atomicAdd(&hist[P(key_of(k))], 1u);
  __syncthreads();
  {  // exclusive scan of the counts: kOrderBins / 1024 consecutive bins per thread
    constexpr int kPer = kOrderBins / 1024;
    uint32_t tot = 0;
    for (int u = 0; u < kPer; ++u) tot += hist[P(tid * kPer + u)];
    uint32_t inc = tot;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - tot;
    for (int w2 = 0; w2 < wave; ++w2) base += wsum[w2];
    for (int u = 0; u < kPer; ++u) { const uint32_t v = hist[P(tid * kPer + u)]; hist[P(tid * kPer + u)] = base; base += v; }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = tid + u * 1024;
    if (k < K) order[hist[P(rkey[u])] + rrank[u]] = k;
  }
  for (int k = tid + 4096; k < K; k += 1024) order[hist[P(key_of(k))] + (uint32_t)tmp[k]] = k;
}

// (declared in obb_rroi_internal.h: the channels-last translation unit uses the same order launch)
int rroi_order_launch(const LevelTable &lv, const int *roi_level_dev, int num_levels, const float *rois_dev, int64_t K,
                      int roi_kind, int *order_ws, void *stream) {
  hipLaunchKernelGGL(rroi_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lv, roi_level_dev, num_levels, rois_dev,
                     (int)K, roi_kind, order_ws, order_ws + K);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

int rroi_check(const char *who, int B, int C, int H, int W, int64_t K, int ph, int pw) {
  if (B < 0 || C < 0 || H < 0 || W < 0 || K < 0)
    return set_err(OBB_EINVAL, "%s: negative dimension", who);
  if (ph <= 0 || pw <= 0) return set_err(OBB_EINVAL, "%s: out size must be positive (%d x %d)", who, ph, pw);
  if ((int64_t)ph * pw > 4096) return set_err(OBB_EINVAL, "%s: out size %d x %d too large", who, ph, pw);
  if (K > 2147483647LL) return set_err(OBB_EINVAL, "%s: too many RoIs", who);
  return OBB_OK;
}

}  // namespace obb

namespace obb {

// channels per block.  One XCD works on one chunk at a time (block_to_work), so a small chunk
// would keep its planes L2 resident (8 channels of a 256 x 256 map = 2 MiB) -- but a block then
// amortises its prologue (frame, sample table) over too little work: measured at configs[2]
// (tools/exp_rroi_cpb.py) forward 81 / 55 / 48 / 96 us and backward 0.33 / 0.30 / 0.34 /
// 0.48 ms for 8 / 16 / 32 / 64 channels per block.
static int rroi_chan_per_block(int C, bool backward) {
  int cpb = backward ? 16 : 32;
  while (cpb > 8 && cpb > C) cpb >>= 1;
  const int64_t g = tuning(OBB_TUNE_RROI_GROUPS, -1);   // test / experiment hook: groups of 8 channels
  if (g > 0 && g <= 64) cpb = (int)g * kChanPerThread;
  return cpb;
}

static int rroi_geometry(const char *who, int C, bool backward, int64_t K, int &cpb, unsigned &chunks,
                         unsigned long long &nblk) {
  cpb = rroi_chan_per_block(C, backward);
  chunks = (unsigned)((C + cpb - 1) / cpb);
  const unsigned tail = chunks & 7u;             // (block_to_work: the last, partial round packs 8 / tail RoIs per group)
  nblk = (unsigned long long)(chunks >> 3) * 8ull * (unsigned long long)K;
  if (tail) nblk += ((unsigned long long)K + 8u / tail - 1) / (8u / tail) * 8ull;
  if (nblk > 2147483647ull) return set_err(OBB_EINVAL, "%s: too many blocks (K=%lld, C=%d)", who, (long long)K, C);
  return OBB_OK;
}

static int rroi_forward_launch_levels(const char *who, const LevelTable &lv, int num_levels, const int *roi_level_dev,
                                      int B, int C, const float *rois_dev, int64_t K, int ph, int pw,
                                      int sample_num, float min_size, int roi_kind, float *out_dev, void *stream,
                                      int *order_ws = nullptr, unsigned char *plan_ws = nullptr) {
  bool x4 = true;   // 16-byte staging loads need 16-byte aligned rows: W % 4 == 0 and an aligned base, on every level
  for (int l = 0; l < num_levels; ++l) {
    int rc = rroi_check(who, B, C, lv.H[l], lv.W[l], K, ph, pw);
    if (rc != OBB_OK) return rc;
    if (K > 0 && C > 0) {
      if (!lv.feat[l]) return set_err(OBB_EINVAL, "%s: null feature map (level %d)", who, l);
      if (lv.H[l] == 0 || lv.W[l] == 0 || B == 0) return set_err(OBB_EINVAL, "%s: empty feature map with %lld RoIs", who, (long long)K);
    }
    x4 = x4 && (lv.W[l] % 4 == 0) && ((reinterpret_cast<uintptr_t>(lv.feat[l]) & 15u) == 0);
  }
  if (K == 0 || C == 0) return OBB_OK;
  if (!rois_dev || !out_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  int cpb;
  unsigned chunks;
  unsigned long long nblk;
  int rc = rroi_geometry(who, C, false, K, cpb, chunks, nblk);
  if (rc != OBB_OK) return rc;
  // with a workspace: the RoIs are visited in Morton order of their centres (one small launch; pays from a few
  // hundred RoIs on: the footprints of RoIs that run together then share lines in an XCD's L2)
  const int *order = nullptr;
  if (order_ws && K >= tuning(OBB_TUNE_RROI_ORDER_MIN, kOrderMinK) && K <= kOrderMaxK) {
    rc = rroi_order_launch(lv, roi_level_dev, num_levels, rois_dev, K, roi_kind, order_ws, stream);
    if (rc != OBB_OK) return rc;
    order = order_ws;
  }
  // with a workspace and enough channels: the per-RoI plan (one wave per RoI), read by all C / 32 blocks of the RoI
  const unsigned char *plan = nullptr;
  if (plan_ws && C >= kPlanMinC && K <= kPlanMaxK && K >= tuning(OBB_TUNE_RROI_PLAN, kPlanMinK)) {
    const unsigned pb = (unsigned)((K + 3) / 4);
    if (x4)
      hipLaunchKernelGGL(rroi_fwd_plan_kernel<true>, dim3(pb), dim3(256), 0, (hipStream_t)stream, lv, roi_level_dev, num_levels, B, C, rois_dev,
                         ph, pw, sample_num, min_size, (int)K, roi_kind, plan_ws);
    else
      hipLaunchKernelGGL(rroi_fwd_plan_kernel<false>, dim3(pb), dim3(256), 0, (hipStream_t)stream, lv, roi_level_dev, num_levels, B, C, rois_dev,
                         ph, pw, sample_num, min_size, (int)K, roi_kind, plan_ws);
    OBB_LAUNCH_CHECK();
    plan = plan_ws;
  }
  if (x4)
    hipLaunchKernelGGL(rroi_forward_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       lv, roi_level_dev, num_levels, B, C, rois_dev, ph, pw, sample_num, min_size, cpb, (int)K, (int)chunks, roi_kind, order, out_dev, plan);
  else
    hipLaunchKernelGGL(rroi_forward_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                       lv, roi_level_dev, num_levels, B, C, rois_dev, ph, pw, sample_num, min_size, cpb, (int)K, (int)chunks, roi_kind, order, out_dev, plan);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

static int rroi_forward_launch(const char *who, const float *feat_dev, int B, int C, int H, int W,
                               const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale,
                               int sample_num, float min_size, int roi_kind, float *out_dev, void *stream) {
  LevelTable lv = {};
  lv.feat[0] = feat_dev; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
  if (K > 0 && C > 0 && !feat_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  return rroi_forward_launch_levels(who, lv, 1, nullptr, B, C, rois_dev, K, ph, pw, sample_num, min_size, roi_kind,
                                    out_dev, stream);
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
  rc = rroi_geometry(who, C, true, K, cpb, chunks, nblk);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(rroi_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     grad_out_dev, B, C, H, W, rois_dev, ph, pw, spatial_scale, sample_num, min_size, cpb, (int)K, (int)chunks, roi_kind, grad_feat_dev, 0, nullptr);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

}  // namespace obb

#if OBB_RROI_ABLATE == 5
extern "C" __attribute__((visibility("default"))) int obb_debug_rroi_stamps(unsigned long long *host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_rroi_stamps), (size_t)words * 8);
}
#endif

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

namespace obb {

GatherLayout gather_layout(int64_t K) {
  GatherLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t k = (size_t)(K > 0 ? K : 1);
  L.plan = take(k * sizeof(RoiPlan));
  L.rstart = take(k * kPlanPix * sizeof(unsigned short));
  L.rentry = take(k * kBwdMaxEntries * sizeof(int2));
  L.lpt = take((size_t)(2 * kLptMax + 16) * sizeof(int));
  L.total = off;
  return L;
}

// the plan pass of the gather formulations (this file's NCHW gather, obb_rroi_nhwc.hip's channels-last one)
int rroi_plan_launch(int B, int H, int W, const float *rois_dev, int64_t K, int ph, int pw, float spatial_scale, int sample_num,
                     float min_size, int roi_kind, RoiPlan *plan, unsigned short *rstart, int2 *rentry, int *lpt, int tiles_x,
                     int tiles, void *stream, const float *tg, float *tgT, int tC) {
  unsigned long long blocks = (unsigned long long)K;
  if (tgT) blocks += (unsigned long long)K * (unsigned)((tC + 63) / 64);
  if (blocks > 2147483647ull) return set_err(OBB_EINVAL, "rroi_plan_launch: too many blocks (K=%lld, C=%d)", (long long)K, tC);
  hipLaunchKernelGGL(rroi_plan_kernel, dim3((unsigned)blocks), dim3(kPlanThreads), 0, (hipStream_t)stream, B, H, W, rois_dev, ph, pw, spatial_scale,
                     sample_num, min_size, roi_kind, plan, rstart, rentry, lpt, tiles_x, tiles, (int)K, tg, tgT, tC);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

}  // namespace obb

extern "C" size_t obb_roi_backward_workspace_bytes(int64_t K) {
  if (K < 0) return 0;
  return obb::gather_layout(K).total;
}

extern "C" int obb_roi_backward_gather_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                           const float *rois_dev, int64_t K, int roi_kind, int ph, int pw,
                                           float spatial_scale, int sample_num, float min_size,
                                           float *grad_feat_dev, void *workspace_dev, size_t workspace_bytes,
                                           void *stream) {
  using namespace obb;
  const char *who = "obb_roi_backward_gather_f32";
  if (roi_kind < 0 || roi_kind > 2) return set_err(OBB_EINVAL, "%s: roi_kind %d (0 rotated, 1 RoIAlign V1, 2 V2-aligned)", who, roi_kind);
  int rc = rroi_check(who, B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (C == 0 || H == 0 || W == 0 || B == 0) return OBB_OK;
  if (!grad_feat_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  hipStream_t st = (hipStream_t)stream;
  const size_t map_elems = (size_t)B * C * H * W;
  if (K == 0) {   // the gradient map is OVERWRITTEN by this entry point: nothing contributes
    OBB_HIP_TRY(hipMemsetAsync(grad_feat_dev, 0, map_elems * sizeof(float), st));
    return OBB_OK;
  }
  if (!grad_out_dev || !rois_dev || !workspace_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  const GatherLayout L = gather_layout(K);
  if (workspace_bytes < L.total)
    return set_err(OBB_EWORKSPACE, "%s: workspace %zu bytes < obb_roi_backward_workspace_bytes(%lld) = %zu", who, workspace_bytes, (long long)K, L.total);
  if ((reinterpret_cast<uintptr_t>(workspace_dev) & 15u) != 0) return set_err(OBB_EINVAL, "%s: workspace must be 16-byte aligned", who);
  if (K > 2147483647LL / 2 || (int64_t)B * C * H * W >= (1ll << 31)) {
    // beyond 32-bit indexing: the atomics path alone (it needs a zeroed map)
    OBB_HIP_TRY(hipMemsetAsync(grad_feat_dev, 0, map_elems * sizeof(float), st));
    return rroi_backward_launch(who, grad_out_dev, B, C, H, W, rois_dev, K, ph, pw, spatial_scale, sample_num, min_size,
                                roi_kind, grad_feat_dev, stream);
  }
  char *ws = static_cast<char *>(workspace_dev);
  RoiPlan *plan = reinterpret_cast<RoiPlan *>(ws + L.plan);
  unsigned short *rstart = reinterpret_cast<unsigned short *>(ws + L.rstart);
  int2 *rentry = reinterpret_cast<int2 *>(ws + L.rentry);
  const int tiles_x = (W + kGTile - 1) / kGTile;
  const unsigned tiles = (unsigned)(tiles_x * ((H + kGTile - 1) / kGTile));
  int *lpt = nullptr;
  if ((long long)tiles * B <= kLptMax) {     // tile counts and the ticket start at zero
    lpt = reinterpret_cast<int *>(ws + L.lpt);
    OBB_HIP_TRY(hipMemsetAsync(lpt, 0, (size_t)(kLptMax + 16) * sizeof(int), st));
  }
  rc = rroi_plan_launch(B, H, W, rois_dev, K, ph, pw, spatial_scale, sample_num, min_size, roi_kind, plan, rstart, rentry, lpt,
                        tiles_x, (int)tiles, stream);
  if (rc != OBB_OK) return rc;
  const int *tile_order = lpt ? lpt + kLptMax + 16 : nullptr;
  // (16 channels per block up to ~1500 RoIs -- configs[2]: 0.068 ms against 0.070 with 32 --, 32 above: a block's list of
  //  the RoIs that meet its tile, built by testing all K plans, is then shared by twice the channels: config-4 shape
  //  0.344 -> 0.318 ms)
  const int gt = (int)tuning(OBB_TUNE_RROI_GATHER_CHAN, K >= 1536 ? 32 : kGChanDefault);
  const int gch = gt >= 32 ? 32 : 16;
  const unsigned chunks = (unsigned)((C + gch - 1) / gch);
  if (tiles > 65535u || B > 65535) return set_err(OBB_EINVAL, "%s: too many map tiles / images", who);
  const dim3 ggrid(chunks, tiles, (unsigned)B);
#define OBB_GATHER(CH, NB)                                                                                              \
  hipLaunchKernelGGL((rroi_gather_kernel<CH, NB>), ggrid, dim3(256), 0, st, grad_out_dev, C, H, W, (int)K, plan, \
                     rstart, rentry, grad_feat_dev, tile_order)
  if (gch == 32) {
    if (ph * pw == 49) OBB_GATHER(32, 49); else OBB_GATHER(32, 0);
  } else {
    if (ph * pw == 49) OBB_GATHER(16, 49); else OBB_GATHER(16, 0);
  }
#undef OBB_GATHER
  OBB_LAUNCH_CHECK();
  // the RoIs the plan could not hold are ADDED by the atomics kernel: its usual geometry (a large RoI's
  // channels spread over 16 blocks), blocks of planned RoIs leave right after the footprint test
  int cpb;
  unsigned achunks;
  unsigned long long nblk;
  rc = rroi_geometry(who, C, true, K, cpb, achunks, nblk);
  if (rc != OBB_OK) return rc;
  hipLaunchKernelGGL(rroi_backward_kernel, dim3((unsigned)nblk), dim3(256), 0, st, grad_out_dev, B, C, H, W, rois_dev, ph, pw,
                     spatial_scale, sample_num, min_size, cpb, (int)K, (int)achunks, roi_kind, grad_feat_dev, 1, plan);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

extern "C" int obb_roi_align_levels_forward_f32(const float *const *feat_devs_host, const int *H_host,
                                                const int *W_host, const float *scale_host, int num_levels,
                                                int B, int C, const float *rois_dev,
                                                const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                                int sample_num, int roi_kind, float min_size,
                                                float *out_dev, void *stream) {
  using namespace obb;
  const char *who = "obb_roi_align_levels_forward_f32";
  if (num_levels < 1 || num_levels > OBB_MAX_LEVELS) return set_err(OBB_EINVAL, "%s: num_levels = %d (1..%d)", who, num_levels, OBB_MAX_LEVELS);
  if (roi_kind < 0 || roi_kind > 2) return set_err(OBB_EINVAL, "%s: bad RoI kind %d", who, roi_kind);
  if (!feat_devs_host || !H_host || !W_host || !scale_host) return set_err(OBB_EINVAL, "%s: null level table", who);
  if (K > 0 && num_levels > 1 && !roi_level_dev) return set_err(OBB_EINVAL, "%s: null level indices", who);
  LevelTable lv = {};
  for (int l = 0; l < num_levels; ++l) {
    lv.feat[l] = feat_devs_host[l]; lv.H[l] = H_host[l]; lv.W[l] = W_host[l]; lv.scale[l] = scale_host[l];
  }
  return rroi_forward_launch_levels(who, lv, num_levels, roi_level_dev, B, C, rois_dev, K, ph, pw, sample_num,
                                    roi_kind == 0 ? min_size : 0.0f, roi_kind, out_dev, stream);
}

namespace obb {
// SingleRoIExtractor.map_roi_levels (single_level.py:54-73) in ONE launch: level = clamp(floor(log2(scale / finest +
// 1e-6)), 0, levels - 1), scale = sqrt(w * h) with w = x2 - x1 + 1 (horizontal RoIs) or the box's own sides (rotated) --
// the same fp32 expression order as the tensor ops it replaces (eleven small launches, ~60 us of host time per call)
__global__ __launch_bounds__(256) void roi_levels_kernel(const float *__restrict__ rois, int64_t K, int roi_kind,
                                                         float finest_scale, int num_levels, int32_t *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const float *r = rois + k * roi_stride(roi_kind);
  const float area = roi_kind == 0 ? r[3] * r[4] : (r[3] - r[1] + 1.0f) * (r[4] - r[2] + 1.0f);
  const float lv = floorf(log2f(sqrtf(area) / finest_scale + 1e-6f));
  const float hi = (float)(num_levels - 1);
  const float c = lv < 0.0f ? 0.0f : (lv > hi ? hi : lv);          // (Tensor.clamp; NaN -> NaN -> (int) 0 below)
  out[k] = c == c ? (int32_t)c : 0;
}
}  // namespace obb

extern "C" int obb_roi_map_levels_i32(const float *rois_dev, int64_t K, int roi_kind, float finest_scale, int num_levels,
                                      int32_t *levels_dev, void *stream) {
  using namespace obb;
  if (K < 0 || roi_kind < 0 || roi_kind > 2 || num_levels < 1 || !(finest_scale > 0.0f))
    return set_err(OBB_EINVAL, "obb_roi_map_levels_i32: bad argument");
  if (K == 0) return OBB_OK;
  if (!rois_dev || !levels_dev) return set_err(OBB_EINVAL, "obb_roi_map_levels_i32: null pointer");
  if (K > 2147483647LL) return set_err(OBB_EINVAL, "obb_roi_map_levels_i32: too many RoIs");
  hipLaunchKernelGGL(roi_levels_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rois_dev, K, roi_kind,
                     finest_scale, num_levels, levels_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}

// [order: 2 K ints | plan records: K x kPlanStride bytes (K <= kPlanMaxK)]
static size_t roi_forward_order_bytes(int64_t K) { return obb::align_up((size_t)(K > 0 ? K : 1) * 2 * sizeof(int32_t), 256); }
extern "C" size_t obb_roi_forward_workspace_bytes(int64_t K) {
  if (K < 0) return 0;
  return roi_forward_order_bytes(K) + (K <= obb::kPlanMaxK ? (size_t)(K > 0 ? K : 1) * obb::kPlanStride : 0);
}

extern "C" int obb_roi_align_forward_ws_f32(const float *const *feat_devs_host, const int *H_host,
                                            const int *W_host, const float *scale_host, int num_levels,
                                            int B, int C, const float *rois_dev,
                                            const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                            int sample_num, int roi_kind, float min_size,
                                            float *out_dev, void *workspace_dev, size_t workspace_bytes,
                                            void *stream) {
  using namespace obb;
  const char *who = "obb_roi_align_forward_ws_f32";
  if (num_levels < 1 || num_levels > OBB_MAX_LEVELS) return set_err(OBB_EINVAL, "%s: num_levels = %d (1..%d)", who, num_levels, OBB_MAX_LEVELS);
  if (roi_kind < 0 || roi_kind > 2) return set_err(OBB_EINVAL, "%s: bad RoI kind %d", who, roi_kind);
  if (!feat_devs_host || !H_host || !W_host || !scale_host) return set_err(OBB_EINVAL, "%s: null level table", who);
  if (K > 0 && num_levels > 1 && !roi_level_dev) return set_err(OBB_EINVAL, "%s: null level indices", who);
  if (workspace_dev) {
    if (workspace_bytes < obb_roi_forward_workspace_bytes(K))
      return set_err(OBB_EWORKSPACE, "%s: workspace %zu bytes < obb_roi_forward_workspace_bytes(%lld) = %zu", who, workspace_bytes,
                     (long long)K, obb_roi_forward_workspace_bytes(K));
    if (reinterpret_cast<uintptr_t>(workspace_dev) & 15u) return set_err(OBB_EINVAL, "%s: workspace must be 16-byte aligned", who);
  }
  LevelTable lv = {};
  for (int l = 0; l < num_levels; ++l) {
    lv.feat[l] = feat_devs_host[l]; lv.H[l] = H_host[l]; lv.W[l] = W_host[l]; lv.scale[l] = scale_host[l];
  }
  return rroi_forward_launch_levels(who, lv, num_levels, roi_level_dev, B, C, rois_dev, K, ph, pw, sample_num,
                                    roi_kind == 0 ? min_size : 0.0f, roi_kind, out_dev, stream,
                                    static_cast<int *>(workspace_dev),
                                    (workspace_dev && K <= kPlanMaxK) ? static_cast<unsigned char *>(workspace_dev) + roi_forward_order_bytes(K) : nullptr);
}

#ifdef OBB_PLAN_STAMPS
extern "C" __attribute__((visibility("default"))) int obb_debug_plan_stamps(unsigned long long *host, int words) {
  if (words > 4096 * 12) words = 4096 * 12;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_plan_stamps), (size_t)words * 8) == hipSuccess ? 0 : -1;
}
#endif
