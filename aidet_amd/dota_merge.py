"""Multi-GPU merge of tiled-inference detections (BASELINE.json configs[4]).

Reference path: every rank pickles its per-patch results,
``collect_results_gpu`` all-gathers the pickles (mmdet/apis/test.py:116-146),
rank 0 writes DOTA text files and the external ``wwtool.mergebypoly_mp`` runs a
per-class polygon NMS on the CPU (mmdet/datasets/dota.py:310-336).

Here the detections stay tensors: one ``all_gather`` of the counts, one
``all_gather`` of the padded ``(Nmax, 8)`` fp32 rows
``[cx, cy, w, h, theta, score, label, image_id]`` over RCCL (xGMI; the payload is
a few MB, i.e. latency bound), then ONE grouped rotated NMS (group = image *
num_classes + label, optional per-class thresholds) on the GPU.  The keep set
equals the single-process result by construction: the gathered rows are the same
multiset and greedy NMS depends on the row order only through score ties (lower
index wins); passing global sequence numbers (``seq``) restores the order a single
process would have produced for the rank-strided patches of
DistributedSampler (distributed_sampler.py:18-25), ties included.
"""
import torch
import torch.distributed as dist

from . import nms_wrapper

COLS = 8  # cx, cy, w, h, theta, score, label, image_id


def shard_patches(num_patches, rank, world_size):
    """Patch indices of one rank: rank, rank+W, rank+2W, ... (the reference's
    DistributedSampler(shuffle=False), distributed_sampler.py:18-25; the padding it
    adds to make shards equal is dropped again by collect_results, test.py:143-145)."""
    return list(range(rank, num_patches, world_size))


def gather_detections(local, group=None):
    """All-gather variable-length detection rows.

    Args:
        local (Tensor): (n_local, C) rows on this rank's device (n_local may be 0).
    Returns:
        list[Tensor]: the rows of every rank, in rank order (on every rank).
    """
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    world = dist.get_world_size(group)
    n = torch.tensor([local.size(0)], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)             # cf. test.py:124
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    pad = local.new_zeros((nmax, local.size(1)))
    pad[:local.size(0)] = local
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)            # cf. test.py:133
    return [p[:c] for p, c in zip(parts, counts)]


def merge_detections(dets, iou_thr, num_classes, class_thrs=None, nms_fn=None):
    """Global per-(image, class) rotated NMS of gathered detections.

    Args:
        dets (Tensor): (N, 8) rows [cx, cy, w, h, theta, score, label, image_id]
            in full-image coordinates.
        iou_thr (float): threshold (reference default for the merge: 0.3 flat,
            dota.py:329-331).
        num_classes (int): number of classes (labels are 0-based).
        class_thrs (sequence, optional): per-class thresholds (dota.py:321-327).
        nms_fn: NMS op ``f(dets(N,6), iou_thr, groups=..., thrs=...) -> (dets, keep)``;
            defaults to the HIP ``rnms``.
    Returns:
        (kept rows (K, 8), keep indices (K,) into ``dets``)
    """
    nms_fn = nms_fn or nms_wrapper.rnms
    if dets.size(0) == 0:
        return dets, torch.zeros(0, dtype=torch.long, device=dets.device)
    labels = dets[:, 6].long()
    groups = dets[:, 7].long() * int(num_classes) + labels
    thrs = None
    if class_thrs is not None:
        thrs = torch.as_tensor(class_thrs, dtype=torch.float32, device=dets.device)[labels]
    _, keep = nms_fn(dets[:, :6].contiguous(), iou_thr, groups=groups, thrs=thrs)
    keep = torch.as_tensor(keep, device=dets.device)
    return dets[keep], keep


def distributed_merge(local, iou_thr, num_classes, class_thrs=None, group=None, nms_fn=None,
                      seq=None):
    """gather_detections + merge_detections.  Every rank returns the same result
    (the reference evaluates on rank 0 only, tools/dota/dota_test.py:157-166).

    Args:
        local (Tensor): (n_local, 8) rows of this rank.
        seq (Tensor, optional): (n_local,) int64 global sequence number of each row
            (e.g. patch_index * 2**20 + row_in_patch).  When given, the gathered
            rows are put back into ascending ``seq`` order, i.e. the order a single
            process visiting patches 0, 1, 2, ... would have produced (the
            interleave of collect_results, test.py:141-145).  Only matters for
            score TIES (lower row index wins); the keep set for distinct scores is
            order independent.
    Returns:
        (kept rows (K, 8), keep indices into the gathered rows, gathered rows)
    """
    if seq is not None:
        rows = torch.cat([local, seq.to(local.dtype).new_zeros((local.size(0), 0))], 1)
        # sequence numbers travel as two float32 columns (hi, lo) to stay exact
        s64 = seq.to(torch.int64)
        hi = (s64 >> 20).to(local.dtype)[:, None]
        lo = (s64 & ((1 << 20) - 1)).to(local.dtype)[:, None]
        rows = torch.cat([local, hi, lo], 1)
        parts = gather_detections(rows, group)
        allrows = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
        key = allrows[:, -2].to(torch.int64) * (1 << 20) + allrows[:, -1].to(torch.int64)
        allrows = allrows[torch.argsort(key, stable=True)][:, :COLS]
    else:
        parts = gather_detections(local, group)
        allrows = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
    kept, keep = merge_detections(allrows, iou_thr, num_classes, class_thrs, nms_fn)
    return kept, keep, allrows
