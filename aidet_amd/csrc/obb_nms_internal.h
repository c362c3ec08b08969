// obb_nms_internal.h -- what obb_tiled.hip needs from obb_nms.hip: the NMS chain enqueued from a generalised box
// source (rows with a pitch; groups / thresholds derived in the prep kernel instead of materialised by a pass of
// their own), without a host synchronisation.  Not part of the C ABI.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace obb {

// Where the prep kernels of an NMS call read a box, its group and its threshold.
//   mode 0  dets (n, pitch) rows; group[] / thrs[] arrays or null (the plain entry points)
//   mode 1  per-patch NMS of the tiled pass (local_patch_nms, dota_merge.py): group = patch[b] * mult + label[b]
//           from the int64 arrays the detector hands over (wraps like Tensor.to(int32)) -- replaces obb_patch_groups_i32
//   mode 2  merged NMS of the tiled pass (merge_detections): rows = [box + score | label | image | ...] with a pitch;
//           group = image * mult + label, thr = thr_table[label] (labels outside [0, mult): the call's threshold)
//           -- replaces obb_merge_rows_split_f32.  dota.py:321-331 are the thresholds the table carries.
// Modes 1 and 2 write the derived group (and threshold) per box to gout / tout for the later stages of the chain.
struct NmsSource {
  const double *dets64;        // float64 horizontal rows [x1, y1, x2, y2, score] (mode 0, OBB_KIND_HBB only; `dets` unused):
                               // the narrow phase runs in double, nms_cpu_kernel<double> (nms_cpu.cpp:62-68)
  uint32_t *klo;               // ... low words of the 64-bit score keys (workspace, n; set by the callee)
  const float *dets;
  int64_t pitch;               // floats per row (>= columns of box + score)
  int mode;
  const int32_t *group;        // mode 0
  const float *thrs;           // mode 0
  const int64_t *a64, *b64;    // mode 1: label, patch index
  int64_t mult;                // modes 1, 2: number of classes
  const float *thr_table;      // mode 2 (may be null: one threshold)
  int allow_grid;              // 1: the caller can repeat the call when the grid route gives up (the synchronous entry points)
  int retain;                  // round 6 (OBB_NMS_RETAIN; obb_nms.hip, NmsRetain): 0 a plain call; 1 SEED the workspace's retained state
                               // (the usual four launches, the tail leaves the state ready); 2 the three-launch call on a seeded workspace
  unsigned long long retain_seq;   // ... the sequence number the workspace's header carries (the host's note of it)
  int32_t *gout;               // modes 1, 2: workspace (n)
  float *tout;                 // mode 2 with a table: workspace (n)
};

struct NmsEnqueued {
  unsigned long long cap = 0, segcap = 0;
  unsigned int ucap = 0;
  bool timed = false, direct = false, fused = false, grid = false;
  int retained = 0;            // what the call did with its workspace's retained state: 0 nothing, 1 seeded it, 2 ran on it (three launches)
};

constexpr int kNmsShards = 64;

// what the host needs after a call: written by the LAST kernel straight into pinned host memory (a device-to-host copy
// of the counters is one more ~5 us launch) or into the caller's device record
struct NmsHostInfo {
  long long kept;
  unsigned long long edges;
  long long rounds;
  unsigned long long cand_total, cand_worst;
  unsigned int heavy_units, light_units;   // scan units over all shards of the unit list
  unsigned int route;        // 0 fused, 1 direct, 2 sorted: which capacity check applies (obb_nms_async_result)
  unsigned int unit_worst;   // fullest shard of the unit list (heavy + light)
};

// Did the call fit the capacities it was launched with?  0 = yes; otherwise a pair capacity that suffices on a retry
// (*why: 1 edges of the fused route, 2 scan units of a segment, 3 candidate pairs / edges of a shard).
//  * fused route: the edge list may use the whole pair capacity (2 * cap entries).
//  * the unit list is 1/8 of the pair capacity, cut into kNmsShards segments: ask for a capacity whose segments hold the
//    fullest one (which tile feeds which segment does not depend on scheduling, and unit_worst segments always fit,
//    so a retry loop terminates).
//  * candidate / edge shards: which of the 64 shards a wave feeds depends on its wave id, i.e. on scheduling: the
//    fullest shard of the retry may differ from this run's, hence the 25 % slack AND the doubling floor -- a caller
//    that loops on OBB_EWORKSPACE (aidet_amd/nms_wrapper.py does) always terminates, because n (n - 1) / 2 pairs per
//    shard is a hard upper bound.
static __host__ __device__ __forceinline__ unsigned long long nms_need_capacity(const NmsHostInfo &h, unsigned long long cap,
                                                                                 unsigned long long segcap, unsigned int ucap,
                                                                                 int *why) {
  if (why) *why = 0;
  if (h.route == 0u) {
    const unsigned long long ecap64 = 2ull * cap;
    const unsigned long long ecap = ecap64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : ecap64;
    if (h.edges > ecap) { if (why) *why = 1; return h.edges / 2 + h.edges / 8 + 1024; }
    return 0;
  }
  if (h.unit_worst > ucap / kNmsShards) {
    unsigned long long need = (unsigned long long)h.unit_worst * kNmsShards * 8;
    need += need / 4 + 1024;
    if (need < cap * 2) need = cap * 2;
    if (why) *why = 2;
    return need;
  }
  if (h.cand_worst > segcap) {
    unsigned long long need = h.cand_worst * kNmsShards;
    need += need / 4 + 1024;
    if (need < cap * 2) need = cap * 2;
    if (why) *why = 3;
    return need;
  }
  return 0;
}

// Validates, lays out `workspace_dev` and enqueues every kernel of one call on `st`; the last kernel writes the
// 64-byte result record (OBB_NMS_RESULT_BYTES) to `result` (device-visible memory).  src.gout / src.tout are set
// by the callee (they live in the workspace).
int nms_enqueue_source(const char *who, NmsSource src, int64_t n, int kind, float iou_thr, int flags_in,
                       int64_t *keep_dev, void *result, void *workspace_dev, size_t workspace_bytes,
                       hipStream_t st, NmsEnqueued &q);

// forgets the calling thread's "skip the grid route for this size class" notes (obb_set_tuning(OBB_TUNE_NMS_GRID, ..))
void nms_grid_hints_clear();

// the keep count of a finished call inside its result record (first 8 bytes: long long kept)
static __device__ __forceinline__ long long nms_record_kept(const void *record) {
  return *reinterpret_cast<const long long *>(record);
}

}  // namespace obb
