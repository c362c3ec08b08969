// obb_rroi_nhwc.hip -- rotated / horizontal RoI-Align on CHANNELS-LAST feature maps (B, H, W, C) for gfx950.
//
// Same operators, values and operation order as obb_rroi.hip (the counterparts of ROIAlignForwardV1 /
// RoIAlignForwardV2, mmdet/ops/roi_align/src/roi_align_kernel.cu:64-117, roi_align_kernel_v2.cu:62-128, and of
// RoIAlignBackwardFeatureV2, roi_align_kernel_v2.cu:179-263); what differs is the map's memory layout.  The reference
// only knows NCHW (roi_align_cuda.cpp:37-42 checks contiguity) -- a PyTorch trunk run in torch.channels_last hands the
// op a (B, C, H, W) tensor whose memory is (B, H, W, C), and until round 5 the wrapper copied it back to NCHW first.
//
// Why the layout matters here.  In NCHW a bilinear tap of one channel is 4 bytes at an arbitrary address: the NCHW
// kernel stages a RoI's footprint rows (~60 B at arbitrary alignment) of every channel plane into LDS and ends at
// 2.4 x the algorithmic bytes in 64-byte sector requests.  In channels-last a tap of ALL channels is one
// contiguous run of 4 C bytes (1 KiB at C = 256):
//   * lane = four consecutive channels (one 16-byte load), a group of NQ lanes = one tap of a slab of 4 NQ channels;
//     a wave-instruction fetches 64 / NQ taps as whole 128-byte lines -- nothing is staged, no footprint is computed;
//   * the sample table (4 tap offsets + 4 weights per sample: the same for every channel) is built once per block in
//     LDS and read back as a broadcast;
//   * the sums are transposed through an LDS tile [channel][bin] -- the block's part of the reference's (K, C, ph, pw)
//     output is one contiguous range -- and leave as 16-byte stores.
// Per output value: w1*v1 + w2*v2 + w3*v3 + w4*v4 summed over the samples in (iy, ix) order, then / count, exactly
// the expressions of rroi_forward_kernel (bit-equal results on the permuted map: tests/test_round5_gpu.py).
#include <cstdlib>

#include "obb_common.h"
#include "obb_geom.h"
#include "obb_rroi_internal.h"

namespace obb {

// OBB_NHWC_STAMPS (tools/build_ablations.sh -> libobbops_nhst.so, tools/exp_nhwc_stamps.py): shader-clock stamps of every
// block's thread 0 -- entry | frame read back | table built | this wave's gathers done | tile complete (barrier) | copy-out
// issued -- and the 100 MHz real-time counter at entry and exit (comparable across CUs: when do the blocks run?).
#ifdef OBB_NHWC_STAMPS
__device__ unsigned long long g_nhwc_stamps[8 * 8192];
#define OBB_NSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_nhwc_stamps[blockIdx.x * 8 + (i)] = (i) >= 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } } while (0)
// (the gather backward: 12 words per tile block -- 0..5 shader clock at entry / list / first batch's scan / fill / stream /
// exit, 6 / 7 the real-time counter at entry / exit, 8 the tile's RoIs, 9 its entries, 10 its batches)
__device__ unsigned long long g_nhwc_gstamps[4096 * 12];
#define OBB_GSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { g_nhwc_gstamps[blockIdx.x * 12 + (i)] = ((i) == 6 || (i) == 7) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } } while (0)
#define OBB_GNOTE(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { g_nhwc_gstamps[blockIdx.x * 12 + (i)] = (unsigned long long)(v); } } while (0)
#else
#define OBB_NSTAMP(i) do { } while (0)
#define OBB_GSTAMP(i) do { } while (0)
#define OBB_GNOTE(i, v) do { } while (0)
#endif

constexpr int kNhwcTab = 256;              // samples of one table round (8 KiB)
constexpr int kNhwcTileBins = 49;          // bins of one output tile when the slab is full (7 x 7: the whole RoI)

// One block = (RoI, slab of CS = 4 NQ channels), NW waves.  lane -> (slot = lane / NQ, quad = lane % NQ): the NW x 64 / NQ
// slots of a block each take a bin at a time; the NQ lanes of a slot hold the slab's channels, four each.
//   rounds   the sample table holds <= 256 samples: bins_per_round = 256 / samples_per_bin bins at a time (7 x 7 bins of
//            2 x 2 samples: one round); a bin of more than 256 samples (adaptive grids of huge RoIs) is summed over
//            several rounds by its one slot
//   tiles    the LDS tile holds CS x TB sums: TB = all bins when they fit (then the copy-out is linear)
// block id -> (RoI id / slabs, slab id % slabs).  (Measured and dropped, LOG round 5: the RoIs in Morton order with one
// contiguous range of the order per XCD -- the order launch costs more than the locality gives here, the XCDs' L2s absorb
// the overlap of the footprints as it is: counter traffic 1.03 x the algorithmic bytes.)
template <int NQ, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void roi_forward_nhwc_kernel(
    const LevelTable lv, const int *__restrict__ roi_level, int num_levels, int B, int C,
    const float *__restrict__ rois, int ph, int pw, int sample_num, float min_size, int K, int slabs,
    int roi_kind, float *__restrict__ out, int out_vec4) {
  constexpr int CS = 4 * NQ;                 // channels of a slab
  constexpr int SL = 64 / NQ;                // slots per wave
  constexpr int kSlots = NW * SL;            // ... per block
  constexpr int kThreads = 64 * NW;
  constexpr int kTileFloats = CS * kNhwcTileBins;
  __shared__ __attribute__((aligned(16))) float otile[kTileFloats];
  __shared__ uint4 s_off[kNhwcTab];          // per sample: byte offsets of the 4 taps' pixels inside the image
  __shared__ float4 s_wgt[kNhwcTab];         // per sample: the 4 bilinear weights (all zero: the sample is outside)
  __shared__ RoiFrame s_frame;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = lane / NQ, quad = lane - slot * NQ, sid = wave * SL + slot;
  OBB_NSTAMP(6);
  OBB_NSTAMP(0);
  const int64_t k = blockIdx.x / (unsigned)slabs;
  const int slab = (int)(blockIdx.x - (unsigned)k * (unsigned)slabs);
  const int lvl_raw = roi_level ? roi_level[k] : 0;
  const bool lvl_ok = lvl_raw >= 0 && lvl_raw < num_levels;
  const int lvl = lvl_ok ? lvl_raw : 0;
  const float *__restrict__ feat = lv.feat[lvl];
  const int H = lv.H[lvl], W = lv.W[lvl];
  // the RoI's frame (trigonometry, bin sizes): ONE wave computes it, the block reads it back (obb_rroi.hip: the other
  // waves leave their issue slots to the gathers of the CU's other blocks)
  if (tid < 64) {
    RoiFrame f0;
    make_frame(rois + k * roi_stride(roi_kind), roi_kind, lv.scale[lvl], ph, pw, sample_num, min_size, f0);
    if (tid == 0) s_frame = f0;
  }
  __syncthreads();
  const RoiFrame f = s_frame;
  OBB_NSTAMP(1);
  const int nb = ph * pw;
  const int c0 = slab * CS, cs = min(CS, C - c0);
  float *__restrict__ obase = out + ((int64_t)k * C + c0) * nb;         // [cs][nb], contiguous
  const long long spb64 = (long long)f.gh * f.gw;
  if (!lvl_ok || f.b < 0 || f.b >= B || spb64 <= 0) {
    // invalid batch / level index: zeros.  Empty sampling grid: the sum of nothing over the count -- 0 (V2, rotated)
    // or 0 / 0 = NaN (V1 divides by gh * gw, roi_align_kernel.cu:113)
    const float v = (!lvl_ok || f.b < 0 || f.b >= B) ? 0.0f : 0.0f / f.cnt;
    for (int it = tid; it < cs * nb; it += kThreads) obase[it] = v;
    return;
  }
  const bool active = 4 * quad < cs;
  const char *__restrict__ img = reinterpret_cast<const char *>(feat + (int64_t)f.b * H * W * C + c0 + 4 * (active ? quad : 0));
  const uint32_t pixb = (uint32_t)C * 4u;      // bytes of a pixel
  const int spb = (int)(spb64 > (1 << 30) ? (1 << 30) : spb64);
  const int qn_max = spb < kNhwcTab ? spb : kNhwcTab;              // samples of a bin per round
  const int bpr = spb <= kNhwcTab ? kNhwcTab / spb : 1;             // bins per round
  const int tb = min(nb, kTileFloats / CS);                          // bins per tile (>= kNhwcTileBins)
  // mean over the samples: a division, or -- a power-of-two count (2 x 2 samples), the usual case -- the exact same
  // value from a multiplication (scaling by a power of two rounds identically)
  const uint32_t cbits = __float_as_uint(f.cnt);
  const bool cnt_pow2 = (cbits & 0x007FFFFFu) == 0u && f.cnt >= 1.0f;
  const float cnt_inv = cnt_pow2 ? 1.0f / f.cnt : 0.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add_sample = [&](const float4 &w, const float4 &p0, const float4 &p1, const float4 &p2, const float4 &p3) {
    float4 v;
    v.x = w.x * p0.x; v.y = w.x * p0.y; v.z = w.x * p0.z; v.w = w.x * p0.w;
    v.x = v.x + w.y * p1.x; v.y = v.y + w.y * p1.y; v.z = v.z + w.y * p1.z; v.w = v.w + w.y * p1.w;
    v.x = v.x + w.z * p2.x; v.y = v.y + w.z * p2.y; v.z = v.z + w.z * p2.z; v.w = v.w + w.z * p2.w;
    v.x = v.x + w.w * p3.x; v.y = v.y + w.w * p3.y; v.z = v.z + w.w * p3.z; v.w = v.w + w.w * p3.w;
    acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
  };
  auto tap = [&](uint32_t off) { return *reinterpret_cast<const float4 *>(img + off); };
  for (int t0 = 0; t0 < nb; t0 += tb) {
    const int t1 = min(nb, t0 + tb), tp = t1 - t0;
    for (int b0 = t0; b0 < t1; b0 += bpr) {
      const int nbr = min(bpr, t1 - b0);
      for (int q0 = 0; q0 < spb; q0 += qn_max) {
        const int qn = min(qn_max, spb - q0);
        // ---- sample table of this round: coordinates, taps and weights ONCE per block (the reference recomputes
        // them for every output element, roi_align_kernel.cu:70-113)
        for (int sidx = tid; sidx < nbr * qn; sidx += kThreads) {
          const int bl = sidx / qn, qq = sidx - bl * qn;
          const int bin = b0 + bl, q = q0 + qq;
          const int bi = bin / pw, bj = bin - bi * pw;
          const int iy = q / f.gw, ix = q - iy * f.gw;
          float x, y, w[4];
          int yl, xl, yh, xh;
          sample_xy(f, bi, bj, iy, ix, x, y);
          const bool live = bilinear_taps_xy(H, W, y, x, yl, xl, yh, xh, w);
          const uint32_t r0 = (uint32_t)yl * (uint32_t)W, r1 = (uint32_t)yh * (uint32_t)W;
          s_off[sidx] = live ? make_uint4((r0 + (uint32_t)xl) * pixb, (r0 + (uint32_t)xh) * pixb, (r1 + (uint32_t)xl) * pixb, (r1 + (uint32_t)xh) * pixb)
                             : make_uint4(0u, 0u, 0u, 0u);
          s_wgt[sidx] = live ? make_float4(w[0], w[1], w[2], w[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        OBB_NSTAMP(2);
        // ---- gathers: a slot takes a bin at a time, 4 samples = 16 loads of 16 bytes in flight per lane
        for (int bl = sid; bl < nbr; bl += kSlots) {
          if (q0 == 0) acc = make_float4(0.f, 0.f, 0.f, 0.f);
          const int e0 = bl * qn;
          int qq = 0;
          if (active) {
            for (; qq + 4 <= qn; qq += 4) {
              uint4 o[4];
              float4 w[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) { o[u] = s_off[e0 + qq + u]; w[u] = s_wgt[e0 + qq + u]; }
              float4 p[4][4];
#pragma unroll
              for (int u = 0; u < 4; ++u) { p[u][0] = tap(o[u].x); p[u][1] = tap(o[u].y); p[u][2] = tap(o[u].z); p[u][3] = tap(o[u].w); }
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (w[u].x != 0.0f) add_sample(w[u], p[u][0], p[u][1], p[u][2], p[u][3]);   // (w1 = 0 <=> the sample is outside the map: it adds nothing)
            }
            for (; qq < qn; ++qq) {
              const uint4 o = s_off[e0 + qq];
              const float4 w = s_wgt[e0 + qq];
              if (w.x != 0.0f) add_sample(w, tap(o.x), tap(o.y), tap(o.z), tap(o.w));
            }
          }
          if (q0 + qn >= spb && active) {
            float4 r;
            if (cnt_pow2) { r.x = acc.x * cnt_inv; r.y = acc.y * cnt_inv; r.z = acc.z * cnt_inv; r.w = acc.w * cnt_inv; }
            else { r.x = acc.x / f.cnt; r.y = acc.y / f.cnt; r.z = acc.z / f.cnt; r.w = acc.w / f.cnt; }
            float *d = otile + (4 * quad) * tp + (b0 + bl - t0);
            d[0] = r.x; d[tp] = r.y; d[2 * tp] = r.z; d[3 * tp] = r.w;
          }
        }
        OBB_NSTAMP(3);
        __syncthreads();      // (the next round rewrites the table; the last one: the tile is complete)
        OBB_NSTAMP(4);
      }
    }
    // ---- copy-out: [channel][bin of the tile] -> out[k][c0 + channel][t0 + bin]
    if (tp == nb) {
      const int n = cs * nb;
      if (out_vec4 && (n & 3) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(otile);
        float4 *d4 = reinterpret_cast<float4 *>(obase);
        for (int i = tid; i < (n >> 2); i += kThreads) d4[i] = s4[i];
      } else {
        for (int i = tid; i < n; i += kThreads) obase[i] = otile[i];
      }
    } else {
      for (int i = tid; i < cs * tp; i += kThreads) {
        const int c = i / tp, j = i - c * tp;
        obase[(int64_t)c * nb + t0 + j] = otile[i];
      }
    }
    OBB_NSTAMP(5);
    OBB_NSTAMP(7);
    __syncthreads();          // (the next tile rewrites otile)
  }
}

// lanes per tap: the slab of a block is 4 NQ channels.  Fewer, wider slabs = longer contiguous runs per tap (1 KiB at
// NQ = 64) and one table per RoI; narrower slabs = a smaller LDS tile (more blocks per CU) and several taps per
// wave-instruction.  OBB_TUNE_RROI_NHWC_NQ overrides (8, 16, 32, 64).
static int nhwc_pick_nq(int C) {
  int nq = 64;
  while (nq > 8 && 4 * (nq >> 1) >= C) nq >>= 1;      // no slab wider than the map has channels
  const int64_t t = tuning(OBB_TUNE_RROI_NHWC_NQ, -1);
  if (t == 8 || t == 16 || t == 32 || t == 64) nq = (int)t;
  return nq;
}

static int roi_forward_nhwc_launch(const char *who, const LevelTable &lv, int num_levels, const int *roi_level_dev,
                                   int B, int C, const float *rois_dev, int64_t K, int ph, int pw, int sample_num,
                                   float min_size, int roi_kind, float *out_dev, void *stream, int *order_ws) {
  for (int l = 0; l < num_levels; ++l) {
    int rc = rroi_check(who, B, C, lv.H[l], lv.W[l], K, ph, pw);
    if (rc != OBB_OK) return rc;
    if (K > 0 && C > 0) {
      if (!lv.feat[l]) return set_err(OBB_EINVAL, "%s: null feature map (level %d)", who, l);
      if (lv.H[l] == 0 || lv.W[l] == 0 || B == 0) return set_err(OBB_EINVAL, "%s: empty feature map with %lld RoIs", who, (long long)K);
      if ((reinterpret_cast<uintptr_t>(lv.feat[l]) & 15u) != 0) return set_err(OBB_EINVAL, "%s: feature map (level %d) must be 16-byte aligned", who, l);
      if ((int64_t)lv.H[l] * lv.W[l] * C >= (1ll << 30))
        return set_err(OBB_EINVAL, "%s: one image of level %d holds %lld floats (the channels-last path addresses < 2^30)", who, l,
                       (long long)((int64_t)lv.H[l] * lv.W[l] * C));
    }
  }
  if (K == 0 || C == 0) return OBB_OK;
  if (C % 4 != 0) return set_err(OBB_EINVAL, "%s: C = %d is not a multiple of 4 (16-byte channel groups)", who, C);
  if (!rois_dev || !out_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  const int nq = nhwc_pick_nq(C);
  const int slabs = (C + 4 * nq - 1) / (4 * nq);
  (void)order_ws;      // (a workspace is accepted for the signature of obb_roi_align_forward_ws_f32; nothing is ordered here)
  const unsigned long long nblk = (unsigned long long)K * (unsigned)slabs;
  if (nblk > 2147483647ull) return set_err(OBB_EINVAL, "%s: too many blocks (K=%lld, C=%d)", who, (long long)K, C);
  const int vec4 = (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0 ? 1 : 0;
  // (more than two RoIs per CU: 8 waves per block -- 2000 RoIs 73.5 us against 80; at 512 RoIs both take 24 us)
  const int nw = tuning(OBB_TUNE_RROI_NHWC_WAVES, K >= 1024 ? 8 : 4) == 8 ? 8 : 4;
#define OBB_NHWC_FWD(NQ, NW)                                                                                               \
  hipLaunchKernelGGL((roi_forward_nhwc_kernel<NQ, NW>), dim3((unsigned)nblk), dim3(64 * NW), 0, (hipStream_t)stream, lv, roi_level_dev, \
                     num_levels, B, C, rois_dev, ph, pw, sample_num, min_size, (int)K, slabs, roi_kind, out_dev, vec4)
  switch (nq + nw) {
    case 64 + 4: OBB_NHWC_FWD(64, 4); break;
    case 64 + 8: OBB_NHWC_FWD(64, 8); break;
    case 32 + 4: OBB_NHWC_FWD(32, 4); break;
    case 32 + 8: OBB_NHWC_FWD(32, 8); break;
    case 16 + 4: OBB_NHWC_FWD(16, 4); break;
    case 16 + 8: OBB_NHWC_FWD(16, 8); break;
    case 8 + 8: OBB_NHWC_FWD(8, 8); break;
    default: OBB_NHWC_FWD(8, 4); break;
  }
#undef OBB_NHWC_FWD
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}


// ==== backward on channels-last maps =========================================================================================
// grad_out (K, C, ph, pw) in the reference's layout -> grad_feat (B, H, W, C).  The gather formulation of obb_rroi.hip with
// the roles of lane and loop exchanged: there a thread owns a PIXEL and walks its tap lists for 16 channels (lanes of a wave
// have lists of different lengths, pixels outside every footprint idle); here a LANE owns four CHANNELS, the (pixel, RoI,
// tap) triple is wave-uniform, and every entry costs the same: one 16-byte load per lane (a contiguous 4 C-byte row of the
// transposed gradient) and four fused multiply-adds.
//   plan + transpose  ONE launch: rroi_plan_kernel's blocks (per RoI: footprint pixel -> (bin, weight / count) of every tap
//              on it, tap order) and, beside their dependent chains, the blocks that transpose grad_out (K, C, nb) ->
//              gT (K, nb, C) in the workspace: a tap then reads one contiguous row
//   gather     one block per 8 x 8 map tile: the RoIs whose footprint meets the tile (all K plans tested, RoI order); per
//              batch of <= 32 of them the list bounds of every (RoI, pixel) pair are fetched at once, a scan per pixel lays
//              the pixels' runs out (RoI order inside = the summation order), the pairs copy their entries into the runs,
//              and every wave streams the runs of its range of pixels (the ranges cut by entries, not by pixels), twelve
//              rows in flight across pixel boundaries, the pixel's sums in four registers per lane; ONE plain 4 C-byte
//              store per pixel, written whether a RoI touched it or not (no zero fill)
//   direct     RoIs the plan cannot hold are added by float atomics (4 C contiguous bytes per tap)
// Summation order per map element: RoI index, then tap number, fused multiply-adds -- the order of rroi_gather_kernel:
// bit-equal to the NCHW gather on the permuted map, reproducible run to run (no atomics for planned RoIs).
// (Round 5 also built the gather WITHOUT the inversion -- per-RoI tap tables filtered by the tile, the hits summed in (RoI,
// tap) order from an LDS stage: tools/attic/obb_rroi_nhwc_taptable_variant.hip.txt.  Its table pass is 10 us cheaper than
// the plan, its gather 35 us slower: a block's filter chain takes 7 us and a staged row 650 cycles of a wave's life.)
typedef float fvec4 __attribute__((ext_vector_type(4)));
constexpr int kNT = 8;                                   // map tile of the gather: 8 x 8 pixels
constexpr int kNPix = kNT * kNT;
constexpr int kNList = 512;                              // RoIs of one tile held in LDS at a time
constexpr int kNBatch = 32;                              // ... whose entries are compacted together (count matrix 32 x 64)
constexpr int kNEnt = 2048;                              // entries of one batch (>= kBwdMaxEntries: a batch of ONE RoI always fits)
constexpr int kNDepth = 12;                              // rows of the transposed gradient a wave has in flight
static_assert(kNEnt >= kBwdMaxEntries, "one RoI's entries fit the batch buffer");

// One block per 8 x 8 map tile.  Per batch of <= 32 listed RoIs: (1) every (RoI, pixel) pair fetches its list bounds --
// all pairs at once, one round trip; (2) a scan per pixel over the RoIs lays the pixels' runs out pixel-major, RoI order
// inside (= the summation order); (3) the pairs copy their entries into the runs as (row of gT, weight); (4) a wave
// streams the runs of its 16 pixels.  Every step's global loads are independent of each other.
template <int NB, int NWV>
__global__ __launch_bounds__(64 * NWV) void roi_gather_nhwc_kernel(
    const float *__restrict__ gT, int C, int H, int W, int K, int nb_rt, const RoiPlan *__restrict__ plan,
    const unsigned short *__restrict__ rstart, const int2 *__restrict__ rentry, float *__restrict__ gfeat) {
  constexpr int kT = 64 * NWV;
  constexpr int kL = kT > kNList ? kT : kNList;            // (a round of kT tested RoIs always fits)
  __shared__ int list[kL];
  __shared__ int4 foot[kL];                                // footprints of the listed RoIs
  __shared__ unsigned int mo[kNBatch][kNPix];              // (RoI of the batch, pixel): entries | offset inside the pixel's run << 16
  __shared__ unsigned short e0s[kNBatch][kNPix];           // ... first entry in the RoI's own list
  __shared__ int pbase[kNPix + 1];                         // first entry of every pixel's run
  __shared__ __attribute__((aligned(16))) int2 ent[kNEnt]; // (row of gT, weight bits)
  __shared__ int wcnt[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = NB ? NB : nb_rt;
  const int tiles_x = (W + kNT - 1) / kNT;
  const int b = (int)blockIdx.y;
  const int tl = (int)blockIdx.x;
  const int tx0 = (tl % tiles_x) * kNT, ty0 = (tl / tiles_x) * kNT;
  const int nq = C >> 2;                                   // 16-byte channel groups of a pixel
  OBB_GSTAMP(6);
  OBB_GSTAMP(0);
  [[maybe_unused]] int st_ent = 0, st_roi = 0, st_bat = 0;
  for (int q0 = 0; q0 < nq; q0 += 64) {                    // (C > 256: slabs of 256 channels, the lists walked again)
    const bool act = q0 + lane < nq;
    const int cq = 4 * (q0 + (act ? lane : 0));
    bool stored = false;                                   // (block-uniform) the tile's pixels hold sums of earlier batches
    // the stream of one batch: wave w owns a contiguous range of the tile's pixels -- the ranges cut so that every wave gets
    // about a quarter of the batch's ENTRIES -- and sums their runs back to back: kNDepth rows in flight whatever pixel they
    // belong to (groups cut at pixel boundaries -- a pixel has ~3 entries -- are a round trip per pixel: 53 us instead of 42)
    auto stream = [&](bool have, bool last) {
      int p0 = (kNPix / NWV) * wave, p1 = p0 + kNPix / NWV;
      if (have) {
        const int total = pbase[kNPix];
        const int mine = pbase[lane];                        // (lane = pixel)
        const uint64_t lo = __ballot(mine * NWV >= total * wave), hi = __ballot(mine * NWV >= total * (wave + 1));
        p0 = wave == 0 ? 0 : (lo ? __ffsll((long long)lo) - 1 : kNPix);
        p1 = wave == NWV - 1 ? kNPix : (hi ? __ffsll((long long)hi) - 1 : kNPix);
      }
      if (p0 >= p1) return;
      int e = have ? __builtin_amdgcn_readfirstlane(pbase[p0]) : 0;
      const int eend = have ? __builtin_amdgcn_readfirstlane(pbase[p1]) : 0;
      int p = p0;
      int pnext = have ? __builtin_amdgcn_readfirstlane(pbase[p0 + 1]) : 0;
      auto pix_ptr = [&](int pp) { return gfeat + (((int64_t)b * H + (ty0 + (pp >> 3))) * W + (tx0 + (pp & 7))) * C + cq; };
      auto inmap = [&](int pp) { return tx0 + (pp & 7) < W && ty0 + (pp >> 3) < H; };
      auto init = [&](int pp) {
        return (stored && act && inmap(pp)) ? *reinterpret_cast<const float4 *>(pix_ptr(pp)) : make_float4(0.f, 0.f, 0.f, 0.f);
      };
      float4 acc = init(p);
      auto advance = [&]() {                               // the current pixel's run is done: store it, next pixel
        if (act && inmap(p)) {                             // (the final sums stream past the L2: the rows of gT stay in it)
          if (last) __builtin_nontemporal_store(fvec4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<fvec4 *>(pix_ptr(p)));
          else *reinterpret_cast<float4 *>(pix_ptr(p)) = acc;
        }
        ++p;
        if (p < p1) {
          acc = init(p);
          pnext = have ? __builtin_amdgcn_readfirstlane(pbase[p + 1]) : 0;
        }
      };
      while (e < eend) {
        const int cnt = min(kNDepth, eend - e);
        const int2 mine = ent[e + min(lane, cnt - 1)];     // (lanes 0 .. kNDepth - 1: the group's entries)
        float4 v[kNDepth];
        float w[kNDepth];
#pragma unroll
        for (int u = 0; u < kNDepth; ++u) {
          const int row_u = __builtin_amdgcn_readlane(mine.x, u);
          w[u] = __int_as_float(__builtin_amdgcn_readlane(mine.y, u));
          v[u] = *reinterpret_cast<const float4 *>(gT + (int64_t)row_u * C + cq);
        }
#pragma unroll
        for (int u = 0; u < kNDepth; ++u) {
          if (u < cnt) {                                   // (wave-uniform)
            while (e + u >= pnext) advance();              // (runs of pixels without entries are empty: stored as they are)
            acc.x = fmaf(v[u].x, w[u], acc.x); acc.y = fmaf(v[u].y, w[u], acc.y);
            acc.z = fmaf(v[u].z, w[u], acc.z); acc.w = fmaf(v[u].w, w[u], acc.w);
          }
        }
        e += cnt;
      }
      while (p < p1) advance();
    };
    for (int r0 = 0; r0 < K;) {
      // ---- the RoIs whose footprint meets this tile, in RoI order (rroi_gather_kernel's list), with their footprints: all K
      // plans are tested, kT at a time, and the list is summed when the next kT might not fit it (a tile of configs[2]
      // lists 4 RoIs, of 2000 RoIs 15: one pass; a pass per 512 TESTED RoIs re-read and re-wrote the tile four times)
      int n = 0;
      for (; r0 < K && n + kT <= kL; r0 += kT) {
        const int r = r0 + tid;
        const RoiPlan p = plan[min(r, K - 1)];
        const bool hit = r < K && p.state == 1 && p.b == b && p.x0 < tx0 + kNT && p.x0 + p.w > tx0 && p.y0 < ty0 + kNT && p.y0 + p.h > ty0;
        const uint64_t mk = __ballot(hit);
        if (lane == 0) wcnt[wave] = __popcll(mk);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < NWV; ++w2) { const int v = wcnt[w2]; off += (w2 < wave) ? v : 0; tot += v; }
        if (hit) {
          const int pos = n + off + __popcll(mk & ((1ull << lane) - 1ull));
          list[pos] = r;
          foot[pos] = make_int4(p.x0, p.y0, p.w, p.h);
        }
        n += tot;
        __syncthreads();
      }
      if (st_bat == 0 && st_roi == 0) OBB_GSTAMP(1);
      st_roi += n;
      int nbr = kNBatch;
      for (int rb0 = 0; rb0 < n;) {
        int cur, total;
        while (true) {
          cur = min(nbr, n - rb0);
          // (1) bounds of every (RoI of the batch, pixel) list
          for (int idx = tid; idx < cur * kNPix; idx += kT) {
            const int rl = idx >> 6, pp = idx & 63;
            const int4 f = foot[rb0 + rl];
            const int px = tx0 + (pp & 7) - f.x, py = ty0 + (pp >> 3) - f.y;
            unsigned int m = 0, e0 = 0;
            if (px >= 0 && px < f.z && py >= 0 && py < f.w) {
              const unsigned short *rs = rstart + (int64_t)list[rb0 + rl] * kPlanPix + (py * f.z + px);
              e0 = rs[0];
              m = (unsigned int)rs[1] - e0;
            }
            mo[rl][pp] = m;
            e0s[rl][pp] = (unsigned short)e0;
          }
          __syncthreads();
          // (2) per pixel: offsets of the RoIs' pieces inside its run; then the runs' first entries
          if (tid < kNPix) {
            unsigned int run = 0;
            for (int rl = 0; rl < cur; ++rl) { const unsigned int m = mo[rl][tid]; mo[rl][tid] = m | (run << 16); run += m; }
            int inc = (int)run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
            pbase[tid] = inc - (int)run;
            if (tid == kNPix - 1) pbase[kNPix] = inc;
          }
          __syncthreads();
          total = pbase[kNPix];
          if (total <= kNEnt || cur == 1) break;
          nbr = max(1, cur >> 1);                           // too many entries for the buffer: half the batch
          __syncthreads();
        }
        if (st_bat == 0) OBB_GSTAMP(2);
        st_ent += total;
        // (3) the pairs copy their entries into the pixels' runs
        for (int idx = tid; idx < cur * kNPix; idx += kT) {
          const int rl = idx >> 6, pp = idx & 63;
          const unsigned int v = mo[rl][pp];
          const int m = (int)(v & 0xFFFFu), off = (int)(v >> 16);
          if (m) {
            const int r = list[rb0 + rl];
            const int2 *src = rentry + (int64_t)r * kBwdMaxEntries + e0s[rl][pp];
            int2 *dst = ent + pbase[pp] + off;
            const int rowb = r * nb;
            for (int t = 0; t < m; ++t) { const int2 en = src[t]; dst[t] = make_int2(rowb + en.x, en.y); }
          }
        }
        __syncthreads();
        if (st_bat == 0) OBB_GSTAMP(3);
        // (4) the runs are summed
        stream(true, r0 >= K && rb0 + cur >= n);
        if (st_bat == 0) OBB_GSTAMP(4);
        ++st_bat;
        stored = true;
        rb0 += cur;
        nbr = kNBatch;
        __syncthreads();      // (the next batch rewrites the buffers)
      }
    }
    if (!stored) stream(false, true);                            // no RoI meets the tile: zeros
  }
  OBB_GSTAMP(5);
  OBB_GSTAMP(7);
  OBB_GNOTE(8, st_roi);
  OBB_GNOTE(9, st_ent);
  OBB_GNOTE(10, st_bat);
}

// RoIs the plan does not hold (state 2: footprint > 2047 pixels, > 256 samples, > 64 bins): every tap adds g * w / count to
// its pixel's 4 C contiguous bytes with float atomics (rroi_backward_direct's expressions).  One block per RoI; blocks of
// planned RoIs leave at the first word.
__global__ __launch_bounds__(256) void roi_backward_nhwc_direct_kernel(
    const float *__restrict__ gT, int B, int C, int H, int W, const float *__restrict__ rois, int ph, int pw, float scale,
    int sample_num, float min_size, int roi_kind, const RoiPlan *__restrict__ plan, float *__restrict__ gfeat) {
  const int64_t k = blockIdx.x;
  if (plan[k].state != 2) return;
  RoiFrame f;
  make_frame(rois + k * roi_stride(roi_kind), roi_kind, scale, ph, pw, sample_num, min_size, f);
  if (f.b < 0 || f.b >= B || (long long)f.gh * f.gw <= 0) return;
  const int nb = ph * pw, nq = C >> 2;
  float *img = gfeat + (int64_t)f.b * H * W * C;
  // item = (bin, channel group); the samples of the bin in order
  for (int64_t it = threadIdx.x; it < (int64_t)nb * nq; it += 256) {
    const int bin = (int)(it / nq), q = (int)(it - (int64_t)bin * nq);
    const int bi = bin / pw, bj = bin - bi * pw;
    const float4 g = *reinterpret_cast<const float4 *>(gT + ((int64_t)k * nb + bin) * C + 4 * q);
    for (int iy = 0; iy < f.gh; ++iy) {
      for (int ix = 0; ix < f.gw; ++ix) {
        float x, y;
        sample_xy(f, bi, bj, iy, ix, x, y);
        int o[4];
        float w[4];
        bilinear_taps(H, W, y, x, o, w);
        if (w[0] == 0.0f && w[1] == 0.0f && w[2] == 0.0f && w[3] == 0.0f) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float *p = img + (int64_t)o[t] * C + 4 * q;
          float d;
          d = g.x * w[t]; d = d / f.cnt; atomicAdd(p + 0, d);
          d = g.y * w[t]; d = d / f.cnt; atomicAdd(p + 1, d);
          d = g.z * w[t]; d = d / f.cnt; atomicAdd(p + 2, d);
          d = g.w * w[t]; d = d / f.cnt; atomicAdd(p + 3, d);
        }
      }
    }
  }
}

struct NhwcBwdLayout { GatherLayout g; size_t gT, total; };
static NhwcBwdLayout nhwc_bwd_layout(int64_t K, int C, int nb) {
  NhwcBwdLayout L;
  L.g = gather_layout(K);
  L.gT = L.g.total;
  L.total = align_up(L.gT + (size_t)(K > 0 ? K : 1) * (size_t)(nb > 0 ? nb : 1) * (size_t)(C > 0 ? C : 1) * sizeof(float), 256);
  return L;
}

}  // namespace obb

#ifdef OBB_NHWC_STAMPS
extern "C" __attribute__((visibility("default"))) int obb_debug_nhwc_gather_stamps(unsigned long long *host, int words) {
  if (words > 4096 * 12) words = 4096 * 12;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_nhwc_gstamps), (size_t)words * 8) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int obb_debug_nhwc_stamps(unsigned long long *host, int words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(obb::g_nhwc_stamps), (size_t)words * 8);
}
#endif

extern "C" int obb_roi_align_forward_nhwc_f32(const float *const *feat_devs_host, const int *H_host,
                                              const int *W_host, const float *scale_host, int num_levels,
                                              int B, int C, const float *rois_dev,
                                              const int32_t *roi_level_dev, int64_t K, int ph, int pw,
                                              int sample_num, int roi_kind, float min_size,
                                              float *out_dev, void *workspace_dev, size_t workspace_bytes,
                                              void *stream) {
  using namespace obb;
  const char *who = "obb_roi_align_forward_nhwc_f32";
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
  return roi_forward_nhwc_launch(who, lv, num_levels, roi_level_dev, B, C, rois_dev, K, ph, pw, sample_num,
                                 roi_kind == 0 ? min_size : 0.0f, roi_kind, out_dev, stream, static_cast<int *>(workspace_dev));
}

extern "C" int obb_rroi_align_forward_nhwc_f32(const float *feat_dev, int B, int C, int H, int W,
                                               const float *rois_dev, int64_t K, int ph, int pw,
                                               float spatial_scale, int sample_num, float min_size,
                                               float *out_dev, void *stream) {
  using namespace obb;
  LevelTable lv = {};
  lv.feat[0] = feat_dev; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
  return roi_forward_nhwc_launch("obb_rroi_align_forward_nhwc_f32", lv, 1, nullptr, B, C, rois_dev, K, ph, pw, sample_num,
                                 min_size, 0, out_dev, stream, nullptr);
}

extern "C" size_t obb_roi_backward_nhwc_workspace_bytes(int64_t K, int C, int ph, int pw) {
  if (K < 0 || C < 0 || ph <= 0 || pw <= 0) return 0;
  return obb::nhwc_bwd_layout(K, C, ph * pw).total;
}

extern "C" int obb_roi_backward_gather_nhwc_f32(const float *grad_out_dev, int B, int C, int H, int W,
                                                const float *rois_dev, int64_t K, int roi_kind, int ph, int pw,
                                                float spatial_scale, int sample_num, float min_size,
                                                float *grad_feat_dev, void *workspace_dev, size_t workspace_bytes,
                                                void *stream) {
  using namespace obb;
  const char *who = "obb_roi_backward_gather_nhwc_f32";
  if (roi_kind < 0 || roi_kind > 2) return set_err(OBB_EINVAL, "%s: roi_kind %d (0 rotated, 1 RoIAlign V1, 2 V2-aligned)", who, roi_kind);
  int rc = rroi_check(who, B, C, H, W, K, ph, pw);
  if (rc != OBB_OK) return rc;
  if (C == 0 || H == 0 || W == 0 || B == 0) return OBB_OK;
  if (!grad_feat_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  if (C % 4 != 0) return set_err(OBB_EINVAL, "%s: C = %d is not a multiple of 4 (16-byte channel groups)", who, C);
  if ((reinterpret_cast<uintptr_t>(grad_feat_dev) & 15u) != 0) return set_err(OBB_EINVAL, "%s: the gradient map must be 16-byte aligned", who);
  hipStream_t st = (hipStream_t)stream;
  const size_t map_elems = (size_t)B * C * H * W;
  if (K == 0) {   // the gradient map is OVERWRITTEN by this entry point: nothing contributes
    OBB_HIP_TRY(hipMemsetAsync(grad_feat_dev, 0, map_elems * sizeof(float), st));
    return OBB_OK;
  }
  if (!grad_out_dev || !rois_dev || !workspace_dev) return set_err(OBB_EINVAL, "%s: null pointer", who);
  const int nb = ph * pw;
  const NhwcBwdLayout L = nhwc_bwd_layout(K, C, nb);
  if (workspace_bytes < L.total)
    return set_err(OBB_EWORKSPACE, "%s: workspace %zu bytes < obb_roi_backward_nhwc_workspace_bytes(%lld, %d, %d, %d) = %zu", who,
                   workspace_bytes, (long long)K, C, ph, pw, L.total);
  if ((reinterpret_cast<uintptr_t>(workspace_dev) & 15u) != 0) return set_err(OBB_EINVAL, "%s: workspace must be 16-byte aligned", who);
  if (K > 65535)
    return set_err(OBB_EINVAL, "%s: K = %lld RoIs (the channels-last gather takes <= 65535)", who, (long long)K);
  const long long tiles_x = (W + kNT - 1) / kNT, tiles_y = (H + kNT - 1) / kNT;
  if (tiles_x * tiles_y > 2147483647LL || B > 65535) return set_err(OBB_EINVAL, "%s: too many map tiles / images", who);
  char *ws = static_cast<char *>(workspace_dev);
  RoiPlan *plan = reinterpret_cast<RoiPlan *>(ws + L.g.plan);
  unsigned short *rstart = reinterpret_cast<unsigned short *>(ws + L.g.rstart);
  int2 *rentry = reinterpret_cast<int2 *>(ws + L.g.rentry);
  float *gT = reinterpret_cast<float *>(ws + L.gT);
  // the plan blocks and the transposition's blocks in ONE launch (rroi_plan_kernel: blocks >= K transpose)
  rc = rroi_plan_launch(B, H, W, rois_dev, K, ph, pw, spatial_scale, sample_num, min_size, roi_kind, plan, rstart, rentry, nullptr, 0, 0,
                        stream, grad_out_dev, gT, C);
  if (rc != OBB_OK) return rc;
  const dim3 ggrid((unsigned)(tiles_x * tiles_y), (unsigned)B);
  // waves per tile (OBB_TUNE_RROI_NHWC_GATHER_WAVES: 4, 8, 16).  The launch ends with its heaviest tile (a few small RoIs
  // put ~900 entries on one: ~19 round trips of 12 rows for each of four waves), but more waves per tile do not help:
  // measured whole call at configs[2] 60.1 / 63.1 / 82.5 us with 4 / 8 / 16 (2000 RoIs: 229 / 261 / 329) -- the bigger
  // blocks halve the resident tiles and lengthen every barrier of the list / bounds / scan / fill chain.
  const int gw = (int)tuning(OBB_TUNE_RROI_NHWC_GATHER_WAVES, 4);
#define OBB_NHWC_GATHER(NBV, WV) hipLaunchKernelGGL((roi_gather_nhwc_kernel<NBV, WV>), ggrid, dim3(64 * WV), 0, st, gT, C, H, W, (int)K, nb, plan, rstart, rentry, grad_feat_dev)
  if (nb == 49) { if (gw == 4) OBB_NHWC_GATHER(49, 4); else if (gw == 8) OBB_NHWC_GATHER(49, 8); else OBB_NHWC_GATHER(49, 16); }
  else { if (gw == 4) OBB_NHWC_GATHER(0, 4); else if (gw == 8) OBB_NHWC_GATHER(0, 8); else OBB_NHWC_GATHER(0, 16); }
#undef OBB_NHWC_GATHER
  OBB_LAUNCH_CHECK();
  hipLaunchKernelGGL(roi_backward_nhwc_direct_kernel, dim3((unsigned)K), dim3(256), 0, st, gT, B, C, H, W, rois_dev, ph, pw, spatial_scale,
                     sample_num, min_size, roi_kind, plan, grad_feat_dev);
  OBB_LAUNCH_CHECK();
  return OBB_OK;
}
