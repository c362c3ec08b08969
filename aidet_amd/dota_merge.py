"""Multi-GPU merge of tiled-inference detections (BASELINE.json configs[4]).

Reference path: every rank pickles its per-patch results,
``collect_results_gpu`` all-gathers the pickles (mmdet/apis/test.py:116-146),
rank 0 writes DOTA text files and the external ``wwtool.mergebypoly_mp`` /
``mergebyrec_mp`` run a per-class NMS on the CPU (mmdet/datasets/dota.py:310-336).

Here the detections stay tensors: ONE ``all_gather_into_tensor`` of the counts, ONE of the
padded fp32 rows over RCCL (xGMI; a few MB, i.e. latency bound), then a grouped NMS on the GPU
(group = image * num_classes + label, optional per-class thresholds).  Two merge layouts,
identical results:

  ``replicated``  every rank runs the one global grouped NMS on all gathered rows (fewest
                  collectives; what the reference's rank 0 does, on every rank)
  ``sharded``     (image, class) groups are independent, so rank r merges the groups with
                  ``group % world == r`` and the keep indices are all-gathered -- two more small
                  collectives, 1/world of the NMS work per rank
  ``routed``      every row travels ONLY to the rank that owns its image (``image % world``): one
                  ``all_gather_into_tensor`` of the send-count vectors, one ``all_to_all_single`` of the
                  rows; every rank then restores the order of, and merges, its own images' rows --
                  1/world of EVERY phase (the two layouts above still receive and re-order all rows on
                  every rank).  The kept rows stay with their owner (the reference writes one result per
                  image, dota.py:310-336) unless the caller asks for them everywhere
                  (``gather_result=True``: one more padded all-gather of the kept rows only).

The keep set equals the single-process result by construction: the gathered rows are the same
multiset, and greedy NMS depends on the row order only through score ties (lower index wins);
passing global sequence numbers (``seq``) restores the order a single process would have
produced for the rank-strided patches of DistributedSampler (distributed_sampler.py:18-25),
ties included.

Row layouts (fp32), chosen by the number of columns:
   8  theta-OBB   cx, cy, w, h, theta, score, label, image_id      -> ``rnms``
  11  polygon     x1, y1, ..., x4, y4, score, label, image_id      -> ``poly_nms`` (Task1 text rows)
   7  horizontal  x1, y1, x2, y2, score, label, image_id           -> ``nms``      (Task2 / mergebyrec)
"""
import ctypes

import numpy as np

import torch
import torch.distributed as dist

from . import _collectives as coll
from . import _lib, nms_wrapper

COLS = 8  # cx, cy, w, h, theta, score, label, image_id
LOCAL = 'local'   # pass as `group` to run a function on this rank alone although a process group exists
_LAYOUT = {8: ('rnms', 6), 11: ('poly_nms', 9), 7: ('nms', 5)}   # columns -> (op, columns of box + score)


_KIND = {'rnms': _lib.KIND_RBOX, 'poly_nms': _lib.KIND_POLY, 'nms': _lib.KIND_HBB}


def _on_gpu(*tensors):
    """True when every tensor lives on one HIP device: the row plumbing then runs as the fused kernels of
    aidet_amd/csrc/obb_merge.hip (bit-identical to the tensor expressions below, which remain the CPU path
    and the specification the GPU tests compare with)."""
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.device == tensors[0].device for t in tensors)


def _i64(t):
    return t.detach().to(torch.int64).contiguous()


def _call(dev, fn, *args):
    with _lib.on_device(dev):
        _lib.check(fn(*args, _lib.current_stream_handle(dev)))


def _row_pitch(rows):
    """(tensor to read, row pitch in elements) of (n, c) fp32 rows: a column slice of wider contiguous rows is read in
    place."""
    if rows.dim() == 2 and rows.size(1) > 0 and rows.stride(1) == 1 and rows.stride(0) >= rows.size(1) and rows.size(0) > 0:
        return rows, rows.stride(0)
    rows = rows.contiguous()
    return rows, rows.size(1)


def gather_rows(rows, index, ncols=None):
    """rows[index][:, :ncols] for (n, c) fp32 rows on the GPU (one streaming kernel; torch's index_select
    takes 0.16 ms for 7.7e5 rows of 32 bytes, this 0.02 ms)."""
    ncols = rows.size(1) if ncols is None else int(ncols)
    if not (_on_gpu(rows, index) and rows.dtype == torch.float32 and rows.dim() == 2 and rows.size(1) > 0):
        return rows[index][:, :ncols]
    rows, ld = _row_pitch(rows)
    index = _i64(index)
    out = rows.new_empty((index.numel(), ncols))
    _call(rows.device, _lib.lib().obb_gather_rows_f32, rows.data_ptr(), rows.size(0), ld,
          index.data_ptr(), index.numel(), ncols, out.data_ptr())
    return out


_class_thr_cache = {}


def _class_thr_tensor(class_thrs, dev):
    """per-class thresholds on the device, uploaded once per distinct list"""
    if isinstance(class_thrs, torch.Tensor):
        return class_thrs.detach().to(device=dev, dtype=torch.float32).contiguous()
    key = (str(dev), tuple(float(v) for v in class_thrs))
    t = _class_thr_cache.get(key)
    if t is None:
        if len(_class_thr_cache) > 64:
            _class_thr_cache.clear()
        t = _class_thr_cache[key] = torch.tensor(key[1], dtype=torch.float32, device=dev)
    return t


ROUTE_MAX_WORLD = 64            # obb_rows_route_f32 / obb_rows_restore_f32
RESTORE_MAX_SLOTS = 1 << 20     # (key_max + 1) * world of obb_rows_restore_f32
_order_flags = {}               # device -> pinned int32[1] the restore kernels report a broken key order in


def _order_flag(dev):
    f = _order_flags.get(dev)
    if f is None:
        f = _order_flags[dev] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return f


def _check_order_flag(dev):
    """After a host synchronisation that follows ``restore_rows`` (the merged NMS ends with one): did a source break
    the promise of non-decreasing keys?"""
    f = _order_flags.get(dev)
    if f is not None and int(f[0]) != 0:
        f[0] = 0
        raise ValueError('order_key: the keys of a rank must be non-decreasing in its row order')


def route_rows(rows, world, order_key=None):
    """rows (n, c) -> (the rows in stable buckets by ``int(rows[:, -1]) % world`` with the key appended as one more
    column, counts int64[world + 1]: rows per bucket and, last, how many keys are smaller than their predecessor).
    On the GPU ``obb_rows_route_f32`` (three launches); the tensor expressions below are the CPU path and the
    specification."""
    n, c = rows.size(0), rows.size(1)
    if (_on_gpu(rows) and rows.dtype == torch.float32 and world <= ROUTE_MAX_WORLD
            and (order_key is None or _on_gpu(rows, order_key))):
        L = _lib.lib()
        rows = rows.contiguous()
        key = _i64(order_key) if order_key is not None else None
        out = rows.new_empty((n, c + 1))
        counts = torch.empty(world + 1, dtype=torch.int64, device=rows.device)
        ws = torch.empty(max(int(L.obb_rows_route_workspace_bytes(n, world)), 16), dtype=torch.uint8, device=rows.device)
        _call(rows.device, L.obb_rows_route_f32, rows.data_ptr(), n, c, c - 1, world,
              key.data_ptr() if key is not None else None, out.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel())
        return out, counts
    key = (order_key.to(rows.dtype) if order_key is not None else rows.new_zeros(n))[:, None]
    dest = rows[:, -1].to(torch.int64) % world
    order = _sort_by_small_key(dest.to(rows.dtype), world)
    bad = (order_key[1:] < order_key[:-1]).sum().reshape(1).to(torch.int64) if order_key is not None and n > 1 \
        else torch.zeros(1, dtype=torch.int64, device=rows.device)
    return torch.cat([rows, key], 1)[order], torch.cat([torch.bincount(dest, minlength=world), bad])


def restore_rows(rows, seg_starts, key_max, out_cols, total):
    """The rows of ``len(seg_starts)`` sources (source r = rows[seg_starts[r]:seg_starts[r + 1]], each in
    non-decreasing key = last column; keys outside [0, key_max] are padding) -> the ``total`` real rows in ascending
    key, equal keys by source, ``out_cols`` columns.  On the GPU ``obb_rows_restore_f32`` (a counting sort over
    (key, source) runs: one memset + three launches; the stable 16-bit argsort + gather it replaces: ~65 us for
    1.6e5 rows); the tensor expressions are the CPU path and the specification."""
    world = len(seg_starts)
    if (_on_gpu(rows) and rows.dtype == torch.float32 and key_max is not None and world <= ROUTE_MAX_WORLD
            and (int(key_max) + 1) * world <= RESTORE_MAX_SLOTS):
        L = _lib.lib()
        rows = rows.contiguous()
        out = rows.new_empty((total, out_cols))
        ws = torch.empty(int(L.obb_rows_restore_workspace_bytes(world, int(key_max))), dtype=torch.uint8, device=rows.device)
        seg = (ctypes.c_int64 * world)(*[int(v) for v in seg_starts])
        _call(rows.device, L.obb_rows_restore_f32, rows.data_ptr(), rows.size(0), rows.size(1), rows.size(1) - 1, seg, world,
              int(key_max), int(out_cols), int(total), out.data_ptr(), _order_flag(rows.device).data_ptr(), ws.data_ptr(),
              ws.numel())
        return out
    small = key_max is not None and int(key_max) < (1 << 15) - 1
    key = rows[:, -1]
    key = key.clamp(max=32767.0).to(torch.int16) if small else key.to(torch.int32)
    order = torch.argsort(key, stable=True)[:total]
    return gather_rows(rows, order, out_cols)


def shard_patches(num_patches, rank, world_size):
    """Patch indices of one rank: rank, rank+W, rank+2W, ... (the reference's
    DistributedSampler(shuffle=False), distributed_sampler.py:18-25; the padding it
    adds to make shards equal is dropped again by collect_results, test.py:143-145)."""
    return list(range(rank, num_patches, world_size))


def _world(group):
    if group is LOCAL:
        return 1, 0
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_detections(local, group=None):
    """All-gather variable-length detection rows.

    Two collectives, one host read: ``all_gather_into_tensor`` of the row counts, ``.tolist()``,
    ``all_gather_into_tensor`` of the rows padded to the largest count (the reference gathers
    pickled bytes the same way, test.py:124-133).

    Args:
        local (Tensor): (n_local, C) rows on this rank's device (n_local may be 0).
    Returns:
        list[Tensor]: the rows of every rank, in rank order (on every rank).
    """
    parts, counts, nmax = _gather_padded(local, group)
    return [parts[r * nmax:r * nmax + c] for r, c in enumerate(counts)]


def _gather_padded(local, group=None, pad_last=0.0):
    """-> (all ranks' rows padded to the largest count, the counts, that count); the last column of
    the padding rows holds ``pad_last``."""
    world, _ = _world(group)
    if world == 1:
        return local, [local.size(0)], local.size(0)
    n = torch.tensor([local.size(0)], dtype=torch.int64, device=local.device)
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    coll.all_gather_into_tensor(counts, n, group=group)             # cf. test.py:124
    counts = counts.tolist()
    nmax = max(max(counts), 1)
    pad = local.new_zeros((nmax, local.size(1)))
    pad[:local.size(0)] = local
    if pad_last != 0.0:
        pad[local.size(0):, -1] = pad_last
    parts = local.new_empty((world * nmax, local.size(1)))
    coll.all_gather_into_tensor(parts, pad, group=group)            # cf. test.py:133
    return parts, counts, nmax


def _split_rows(dets, num_classes):
    cols = dets.size(1)
    if cols not in _LAYOUT:
        raise ValueError('detection rows must have 8 (theta-OBB), 11 (polygon) or 7 (horizontal) '
                         'columns, got {}'.format(cols))
    op, nbox = _LAYOUT[cols]
    labels = dets[:, nbox].long()
    groups = dets[:, nbox + 1].long() * int(num_classes) + labels
    return op, dets[:, :nbox].contiguous(), labels, groups


def merge_detections(dets, iou_thr, num_classes, class_thrs=None, nms_fn=None, strict=False,
                     only_groups=None):
    """Global per-(image, class) NMS of gathered detections.

    Args:
        dets (Tensor): (N, 8 | 11 | 7) rows in full-image coordinates (module docstring).
        iou_thr (float): threshold (reference default for the merge: 0.3 flat,
            dota.py:329-331).
        num_classes (int): number of classes (labels are 0-based).
        class_thrs (sequence, optional): per-class thresholds (dota.py:320-327,
            ``dota_io.class_thresholds``).
        nms_fn: NMS op ``f(dets, iou_thr, groups=..., thrs=...) -> (dets, keep)``;
            defaults to the HIP op of the row layout.
        strict (bool): suppress on ``>`` instead of ``>=``.  The DOTA devkit's merge that the
            reference delegates to [external] keeps ``ovr <= thresh``, i.e. it is strict; the
            in-tree CPU NMS (nms_cpu.cpp:56) is not.  Default False = the in-tree rule.
        only_groups ((world, rank), optional): merge only the groups with
            ``group % world == rank`` (the sharded layout).
    Returns:
        (kept rows, keep indices (K,) into ``dets``, ascending)
    """
    if dets.size(0) == 0:
        return dets, torch.zeros(0, dtype=torch.long, device=dets.device)
    if dets.size(1) == 7 and not strict and class_thrs is not None and not all(float(v) > 0.0 for v in class_thrs):
        # horizontal rows follow nms_cpu.cpp to the letter, where a threshold <= 0 under `>=` suppresses every later
        # box of the group, disjoint ones included: refused like nms_wrapper.nms refuses it
        raise ValueError('merge_detections: per-class thresholds of horizontal rows must be positive under the `>=` rule')
    if (nms_fn is None and _on_gpu(dets) and dets.dtype == torch.float32 and dets.size(1) in _LAYOUT
            and float(iou_thr) > 0.0):
        # fused: one kernel splits the rows into boxes | int32 groups | thresholds, the grouped NMS reads
        # them as they are, one kernel gathers the kept rows
        rows, ld0 = _row_pitch(dets)          # (a column slice of wider rows is read in place)
        cols, dev, L = rows.size(1), rows.device, _lib.lib()
        cthr = None
        if class_thrs is not None:
            cthr = _class_thr_tensor(class_thrs, dev)
            assert cthr.numel() >= int(num_classes), 'one threshold per class'

        def split(r, want_boxes):
            r, ld = (rows, ld0) if r is rows else _row_pitch(r)
            m = r.size(0)
            bx = r.new_empty((m, cols - 2)) if want_boxes else None
            gr = torch.empty(m, dtype=torch.int32, device=dev)
            th = torch.empty(m, dtype=torch.float32, device=dev) if (want_boxes and cthr is not None) else None
            _call(dev, L.obb_merge_rows_split_ld_f32, r.data_ptr(), m, cols, ld, int(num_classes),
                  None if cthr is None else cthr.data_ptr(), float(iou_thr), None if bx is None else bx.data_ptr(),
                  gr.data_ptr(), None if th is None else th.data_ptr())
            return bx, gr, th

        sel, mine = None, rows
        if only_groups is not None:      # the sharded layout: this rank's (image, class) groups only
            world, rank = only_groups
            sel = (split(rows, False)[1] % world == rank).nonzero(as_tuple=True)[0]
            if sel.numel() == 0:
                return dets[:0], sel
            mine = gather_rows(rows, sel)
        boxes, groups, thrs = split(mine, True)
        keep, _ = nms_wrapper._run_nms(boxes, _KIND[_LAYOUT[cols][0]], float(iou_thr), groups, thrs, strict)
        if sel is not None:
            keep = sel[keep]
        return gather_rows(rows, keep), keep
    op, boxes, labels, groups = _split_rows(dets, num_classes)
    thrs = None
    if class_thrs is not None:
        thrs = torch.as_tensor(class_thrs, dtype=torch.float32, device=dets.device)[labels]
    kw = {'strict': True} if strict else {}
    sel = None
    if only_groups is not None:
        world, rank = only_groups
        sel = (groups % world == rank).nonzero(as_tuple=True)[0]
        boxes, groups = boxes[sel], groups[sel]
        thrs = None if thrs is None else thrs[sel]
        if sel.numel() == 0:
            return dets[:0], sel
    nms_fn = nms_fn or getattr(nms_wrapper, op)
    _, keep = nms_fn(boxes, iou_thr, groups=groups, thrs=thrs, **kw)
    keep = torch.as_tensor(keep, device=dets.device)
    if sel is not None:
        keep = sel[keep]
    return dets[keep], keep


def _restore_order(parts, ncols):
    """rows of all ranks -> ascending global sequence number (two trailing fp32 columns)."""
    allrows = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
    key = allrows[:, -2].to(torch.int64) * (1 << 20) + allrows[:, -1].to(torch.int64)
    return allrows[torch.argsort(key, stable=True)][:, :ncols]


_PAD_KEY = float((1 << 24) - 1)      # key of the padding rows of the gather: sorts behind every real key


def _restore_by_key(padded, total, ncols, key_max=None, world=1):
    """The padded rows of all ranks (``_gather_padded`` with ``_PAD_KEY``: ``world`` equal blocks, real rows first), a
    small integer key in the last column, every rank's rows in non-decreasing key -> the ``total`` real rows in
    ascending key, rank order inside a key (``restore_rows``), without the key column."""
    nmax = padded.size(0) // max(world, 1)
    kmax = key_max if key_max is not None else None
    return restore_rows(padded, [r * nmax for r in range(max(world, 1))], kmax, ncols, total)


def _sort_by_small_key(key, key_max):
    """stable argsort of small non-negative integer keys (16-bit sort when they fit)."""
    small = key_max is not None and int(key_max) < (1 << 15) - 1
    k = key.clamp(max=32767.0).to(torch.int16) if small else key.to(torch.int32)
    return torch.argsort(k, stable=True)


def routed_merge(local, iou_thr, num_classes, class_thrs=None, group=None, nms_fn=None, strict=False, mark=None,
                 order_key=None, order_key_max=None, gather_result=True):
    """The ``routed`` layout (module docstring): rows -> the owner of their image (``image % world``) by
    all_to_all, order restored and merged there.

    Args:
        local (Tensor): (n_local, 8 | 11 | 7) rows of this rank; the LAST column is the image id.
        order_key (Tensor, optional): (n_local,) small non-negative integers (< 2**24; e.g. the patch index), every
            key value on ONE rank and non-decreasing in this rank's row order: the single-process order is ascending
            key, rank-local order inside a key.  Without it rows of different ranks keep rank order (only score
            TIES between rows of different ranks can then differ from a single-process run).
        gather_result (bool): True -> every rank returns ALL kept rows, in single-process order (what the other
            layouts return); False -> the kept rows of this rank's images only.
    Returns:
        (kept rows, keep indices into this rank's routed rows, this rank's routed rows)
    """
    mark = mark or (lambda name: None)
    world, rank = _world(group)
    ncols = local.size(1)
    dev = local.device
    if world > 1:
        send, counts = route_rows(local, world, order_key)
        mat = torch.empty(world * (world + 1), dtype=torch.int64, device=dev)
        coll.all_gather_into_tensor(mat, counts, group=group)
        mat = mat.view(world, world + 1).tolist()                    # mat[src][dst] | unordered keys; the one host read
        if any(row[world] for row in mat):
            raise ValueError('order_key: the keys of a rank must be non-decreasing in its row order')
        in_splits, out_splits = mat[rank][:world], [mat[src][rank] for src in range(world)]
        recv = send.new_empty((sum(out_splits), ncols + 1))
        coll.all_to_all_single(recv, send, out_splits, in_splits, group=group)
        # rows of one source arrive together and in order; patches interleave over the sources
        mine = recv
        if order_key is not None and mine.size(0) > 0:
            starts = [0]
            for c in out_splits[:-1]:
                starts.append(starts[-1] + c)
            mine = restore_rows(recv, starts, order_key_max, ncols + 1, recv.size(0))
    else:
        key = (order_key.to(local.dtype) if order_key is not None else local.new_zeros(local.size(0)))[:, None]
        mine = torch.cat([local, key], 1)
    mark('gather')
    allrows = mine[:, :ncols]                 # (read in place: the kernels take the row pitch)
    kept, keep = merge_detections(allrows, iou_thr, num_classes, class_thrs, nms_fn, strict)
    if mine.is_cuda:
        _check_order_flag(dev)
    if gather_result and world > 1:
        kk = gather_rows(mine, keep) if kept.size(0) else mine[:0]      # (kept rows with their key column)
        padded, counts, _ = _gather_padded(kk, group, _PAD_KEY)
        if order_key is not None:
            kept = _restore_by_key(padded, sum(counts), ncols, order_key_max, world)
        else:
            nmax = padded.size(0) // world
            kept = torch.cat([padded[r * nmax:r * nmax + c] for r, c in enumerate(counts)], 0)[:, :ncols]
    mark('merge')
    return kept, keep, allrows


def distributed_merge(local, iou_thr, num_classes, class_thrs=None, group=None, nms_fn=None,
                      seq=None, layout='replicated', strict=False, mark=None, order_key=None,
                      order_key_max=None, gather_result=True):
    """gather_detections + merge_detections.  Every rank returns the same result
    (the reference evaluates on rank 0 only, tools/dota/dota_test.py:157-166).

    Args:
        local (Tensor): (n_local, 8 | 11 | 7) rows of this rank.
        seq (Tensor, optional): (n_local,) int64 global sequence number of each row
            (e.g. patch_index * 2**20 + row_in_patch).  When given, the gathered
            rows are put back into ascending ``seq`` order, i.e. the order a single
            process visiting patches 0, 1, 2, ... would have produced (the
            interleave of collect_results, test.py:141-145).  Only matters for
            score TIES (lower row index wins); the keep set for distinct scores is
            order independent.
        order_key (Tensor, optional): (n_local,) small non-negative integers (< 2**24), e.g. the patch
            index, with the same purpose as ``seq`` and one promise that makes restoring the order
            cheap: every key value lives on ONE rank.  The single-process order is then ascending
            key, rank-local order inside a key.  Takes precedence over ``seq``.
        order_key_max (int, optional): an upper bound of the keys (e.g. the number of patches); below
            2**15 the sort runs on 16-bit keys.
        layout: 'replicated' | 'sharded' | 'routed' (module docstring; 'routed' = ``routed_merge``: its keep
            indices / gathered rows are this rank's routed rows, not all rows).
        gather_result (bool): 'routed' only -- False leaves the kept rows with the rank that owns their image.
        mark (callable, optional): ``mark('gather')`` / ``mark('merge')`` are called after the
            respective phase (e.g. to record HIP events).
    Returns:
        (kept rows, keep indices into the gathered rows, gathered rows)
    """
    if layout not in ('replicated', 'sharded', 'routed'):
        raise ValueError('layout must be "replicated", "sharded" or "routed"')
    mark = mark or (lambda name: None)
    ncols = local.size(1)
    if _world(group)[0] == 1:
        seq = order_key = None      # one rank: the local rows ARE the single-process order
    if layout == 'routed':
        if seq is not None and order_key is None:
            raise ValueError("layout='routed' restores the order from `order_key` (small integers), not from `seq`")
        return routed_merge(local, iou_thr, num_classes, class_thrs, group, nms_fn, strict, mark, order_key,
                            order_key_max, gather_result)
    if order_key is not None:
        key = order_key.to(local.dtype)[:, None]         # (exact in fp32 below 2**24)
        padded, counts, _ = _gather_padded(torch.cat([local, key], 1), group, _PAD_KEY)
        allrows = _restore_by_key(padded, sum(counts), ncols, order_key_max, len(counts))
    elif seq is not None:
        # sequence numbers travel as two float32 columns (hi, lo) to stay exact
        s64 = seq.to(torch.int64)
        hi = (s64 >> 20).to(local.dtype)[:, None]
        lo = (s64 & ((1 << 20) - 1)).to(local.dtype)[:, None]
        allrows = _restore_order(gather_detections(torch.cat([local, hi, lo], 1), group), ncols)
    else:
        parts = gather_detections(local, group)
        allrows = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
    mark('gather')
    world, rank = _world(group)
    if layout == 'sharded' and world > 1:
        _, mine = merge_detections(allrows, iou_thr, num_classes, class_thrs, nms_fn, strict,
                                   only_groups=(world, rank))
        # every row belongs to exactly one rank's groups: the keep set is the union, exchanged as ONE
        # all-reduce of a byte mask (an all-gather of index lists needs two collectives and a sort)
        mask = torch.zeros(allrows.size(0), dtype=torch.uint8, device=allrows.device)
        mask[mine] = 1
        coll.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
        keep = mask.nonzero(as_tuple=True)[0]
        kept = gather_rows(allrows, keep)
    else:
        kept, keep = merge_detections(allrows, iou_thr, num_classes, class_thrs, nms_fn, strict)
    if order_key is not None and allrows.is_cuda:
        _check_order_flag(allrows.device)
    mark('merge')
    return kept, keep, allrows


def local_patch_nms(dets, labels, patch_index, iou_thr, num_classes, nms_fn=None):
    """The per-patch, per-class NMS every rank applies to its own detections before the
    exchange (what the detector's ``multiclass_*nms`` does image by image,
    rbbox_head.py:287-295) -- all patches of the rank in ONE grouped launch
    (group = patch * num_classes + label).

    Args:
        dets (Tensor): (n, 6) theta-OBBs + score, patch coordinates.
        labels, patch_index (Tensor): (n,) class and global patch number of each row.
    Returns:
        keep indices (ascending) into ``dets``.
    """
    if dets.size(0) == 0:
        return torch.zeros(0, dtype=torch.long, device=dets.device)
    if nms_fn is None and _on_gpu(dets, labels, patch_index) and dets.dtype == torch.float32:
        d, lab, pid = dets.contiguous(), _i64(labels), _i64(patch_index)
        groups = torch.empty(d.size(0), dtype=torch.int32, device=d.device)
        _call(d.device, _lib.lib().obb_patch_groups_i32, lab.data_ptr(), pid.data_ptr(), d.size(0),
              int(num_classes), groups.data_ptr())
        return nms_wrapper._run_nms(d, _lib.KIND_RBOX, float(iou_thr), groups)[0]
    nms_fn = nms_fn or nms_wrapper.rnms
    _, keep = nms_fn(dets, iou_thr, groups=patch_index.long() * int(num_classes) + labels.long())
    return torch.as_tensor(keep, device=dets.device)


def patch_rows_to_image(dets, labels, patch_index, patch_xy, patch_image, rate=1.0, keep=None, with_patch=False):
    """theta-OBB rows of many patches -> (n, 8) merge rows in image coordinates, vectorised
    (``dota_io.patch_rows_to_image`` does one patch from its file name).

    Args:
        dets (Tensor): (n, 6) cx, cy, w, h, theta, score in patch pixels.
        patch_index (Tensor): (n,) global patch number of each row.
        patch_xy (Tensor): (P, 2) left/top of every patch; patch_image (P,) its image id.
        keep (Tensor, optional): (k,) row indices -- the result is that of ``dets[keep]``,
            ``labels[keep]``, ``patch_index[keep]`` (on the GPU the selection is never materialised).
        with_patch (bool): also return ``patch_index[keep]`` (int64; written by the same kernel) -- the order key
            of the multi-rank merge.
    """
    if (_on_gpu(dets, labels, patch_index, patch_xy, patch_image) and (keep is None or _on_gpu(dets, keep))
            and dets.dtype == torch.float32 and float(rate) > 0.0):
        d, lab, pid = dets.contiguous(), _i64(labels), _i64(patch_index)
        pxy = patch_xy.detach().to(torch.float32).contiguous()
        pim = _i64(patch_image)
        kk = None if keep is None else _i64(keep)
        k = d.size(0) if kk is None else kk.numel()
        out = d.new_empty((k, 8))
        pof = torch.empty(k, dtype=torch.int64, device=d.device) if with_patch else None
        _call(d.device, _lib.lib().obb_patch_rows_to_image_keyed_f32, d.data_ptr(), lab.data_ptr(), pid.data_ptr(),
              d.size(0), None if kk is None else kk.data_ptr(), k, pxy.data_ptr(), pim.data_ptr(), pim.numel(),
              float(rate), out.data_ptr(), None if pof is None else pof.data_ptr())
        return (out, pof) if with_patch else out
    if keep is not None:
        dets, labels, patch_index = dets[keep], labels[keep], patch_index[keep]
    off = patch_xy[patch_index]
    out = dets.new_empty((dets.size(0), 8))
    out[:, 0:2] = (dets[:, 0:2] + off) / rate
    out[:, 2:4] = dets[:, 2:4] / rate
    out[:, 4:6] = dets[:, 4:6]
    out[:, 6] = labels.to(out.dtype)
    out[:, 7] = patch_image[patch_index].to(out.dtype)
    return (out, patch_index.to(torch.int64)) if with_patch else out


# ---------------------------------------------------------------------------------------------------------------
# Fused phases (round 4): ``obb_tiled_local_f32`` / ``obb_tiled_merge_f32`` enqueue a whole rank-side phase each --
# the group ids and per-class thresholds are derived inside the NMS's first kernel, the kernels behind an NMS read
# its keep count from the result record on the device -- so a pass has ONE host read per exchange (the count
# matrix the collectives need anyway) and one at its end (the merged keep count) instead of a wrapper round trip
# per kernel.  The step-by-step functions above stay: they are the CPU path of the gloo tests, the specification the
# GPU tests compare the fused path with, and the route of the ``sharded`` layout.
# ---------------------------------------------------------------------------------------------------------------
TILED_REPEAT = -(1 << 62)    # K word of a phase whose NMS must be repeated as it is (the grid route gave up)
TILED_ABORT = -(1 << 62) - 1  # K word a rank publishes INSTEAD of running its phase: it cannot go on, every rank raises together
phase_repeats = 0            # phases this process repeated for that reason (tests read it)
FUSED = True                 # module switch (tests force the step-by-step path with it)
_phase_ws = {}               # (device index, stream) -> uint8 workspace shared by the phases (stream-ordered)
_phase_capacity = {}         # (phase, bit length of n) -> pair capacity the last overflowing call needed


def _phase_workspace(dev, nbytes):
    key = (dev.index, _lib.current_stream_handle(dev))
    ws = _phase_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _phase_ws[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    return ws


def _capacity_error(what, n, nbytes, capacity):
    return _lib.ObbOpsError('%s on %d rows needs a %.1f GiB pair workspace (capacity %d pairs): more than '
                            'nms_wrapper.MAX_WORKSPACE_BYTES' % (what, n, nbytes / 2.0 ** 30, capacity))


def tiled_local_phase(dets, labels, patch_index, patch_xy, patch_image, num_classes, iou_thr, world=1, rate=1.0,
                      capacity=None):
    """``obb_tiled_local_f32``: per-patch, per-class rotated NMS of this rank's detections -> the kept rows in image
    coordinates with their patch index as a 9th column, in keep order (``world`` 1) or bucketed by destination rank
    ``image % world``.  Nothing is synchronised.

    Returns (rows (n, 9) buffer of which the first K rows are valid, counts int64[world + 2 + 8] on the device:
    rows per destination | order violations | K, or -(pair capacity needed) after an overflow, or TILED_REPEAT when the
    NMS's grid route gave the scene up (pass the record to ``obb_tiled_result`` -- it answers OBB_ERETRY and notes the size
    class for this thread -- and call again as it is) | the NMS record).
    """
    L = _lib.lib()
    dev, n = dets.device, dets.size(0)
    d, lab, pid = dets.contiguous(), _i64(labels), _i64(patch_index)
    pxy = patch_xy.detach().to(torch.float32).contiguous()
    pim = _i64(patch_image)
    rows = d.new_empty((n, 9))
    counts = torch.empty(world + 2 + _lib.NMS_RESULT_BYTES // 8, dtype=torch.int64, device=dev)
    if capacity is None:
        capacity = _phase_capacity.get((0, int(n).bit_length()), 0)
    nbytes = int(L.obb_tiled_local_workspace_bytes(n, world, capacity))
    if nbytes > nms_wrapper.MAX_WORKSPACE_BYTES:
        raise _capacity_error('the per-patch NMS', n, nbytes, capacity)
    ws = _phase_workspace(dev, nbytes)
    _call(dev, L.obb_tiled_local_f32, d.data_ptr(), lab.data_ptr(), pid.data_ptr(), n, pxy.data_ptr(), pim.data_ptr(),
          pim.numel(), int(num_classes), float(rate), float(iou_thr), int(world), rows.data_ptr(), counts.data_ptr(),
          ws.data_ptr(), nbytes)
    counts.phase_nbytes = nbytes           # (the workspace size this phase RAN with: obb_tiled_result checks the record against it)
    return rows, counts


def _note_capacity(phase, n, need):
    _phase_capacity[(phase, int(n).bit_length())] = int(need)


def tiled_merge_phase(recv, seg_starts, key_max, total, num_classes, iou_thr, class_thrs=None, strict=False, cols=8,
                      out_cols=None):
    """``obb_tiled_merge_f32``: the rows of ``len(seg_starts)`` sources (``recv`` (m, ld), columns = ``cols`` merge
    columns [+ ... + key]) -> order restored by the key column (``key_max`` None: used as they are) -> merged
    per-(image, class) NMS -> kept rows.  ONE host read (the result record) at the end; an overflow of the pair lists
    repeats the phase with the capacity the library names.

    Returns (kept rows (K, out_cols), keep indices (K,) into the ordered rows, ordered rows (total, ld))."""
    L = _lib.lib()
    dev = recv.device
    recv, ld = _row_pitch(recv)
    m, world = recv.size(0), len(seg_starts)
    out_cols = cols if out_cols is None else int(out_cols)
    if total == 0:
        return recv.new_empty((0, out_cols)), torch.zeros(0, dtype=torch.int64, device=dev), recv[:0]
    kmax = -1 if key_max is None else int(key_max)
    cthr = None
    if class_thrs is not None:
        cthr = _class_thr_tensor(class_thrs, dev)
        assert cthr.numel() >= int(num_classes), 'one threshold per class'
    ordered = recv.new_empty((total, ld)) if kmax >= 0 else None
    kept = recv.new_empty((total, out_cols))
    keep = torch.empty(total, dtype=torch.int64, device=dev)
    record = torch.empty(_lib.NMS_RESULT_BYTES // 8, dtype=torch.int64, device=dev)
    seg = (ctypes.c_int64 * world)(*[int(v) for v in seg_starts])
    info = (ctypes.c_int64 * 4)()
    capacity = _phase_capacity.get((1, int(total).bit_length()), 0)
    retries = 0
    while True:
        nbytes = int(L.obb_tiled_merge_workspace_bytes(total, cols, world, kmax, capacity))
        if nbytes == 0:
            raise _lib.ObbOpsError('obb_tiled_merge_workspace_bytes failed: ' + L.obb_last_error().decode())
        if nbytes > nms_wrapper.MAX_WORKSPACE_BYTES:
            raise _capacity_error('the merged NMS', total, nbytes, capacity)
        ws = _phase_workspace(dev, nbytes)
        _call(dev, L.obb_tiled_merge_f32, recv.data_ptr(), m, cols, ld, seg, world, kmax, int(total), int(num_classes),
              None if cthr is None else cthr.data_ptr(), float(iou_thr), _lib.NMS_STRICT if strict else 0,
              None if ordered is None else ordered.data_ptr(), kept.data_ptr(), out_cols, keep.data_ptr(),
              record.data_ptr(), _order_flag(dev).data_ptr() if kmax >= 0 else None, ws.data_ptr(), nbytes)
        with _lib.on_device(dev):
            rec = record.cpu().numpy()                                   # the phase's one host read
        rc = L.obb_tiled_result(rec.ctypes.data, 1, int(total), cols, world, kmax, nbytes, info)
        if rc == _lib.OBB_ERETRY:             # the NMS's grid route gave up on this scene: once more, as it is
            global phase_repeats
            phase_repeats += 1
            retries += 1
            if retries > 3:
                raise _lib.ObbOpsError('the merge phase asks to be repeated again and again: ' + L.obb_last_error().decode())
            continue
        if rc != _lib.OBB_EWORKSPACE:
            _lib.check(rc)
            break
        capacity = max(int(info[3]), 2 * capacity)
        _note_capacity(1, total, capacity)
    k = int(info[0])
    return kept[:k], keep[:k], (ordered if ordered is not None else recv)


def _fusable(dets, labels, patch_index, patch_xy, patch_image, layout, nms_fn, local_thr, merge_thr, class_thrs):
    return (FUSED and nms_fn is None and layout in ('replicated', 'routed') and dets.dim() == 2 and dets.size(1) == 6
            and dets.dtype == torch.float32 and _on_gpu(dets, labels, patch_index, patch_xy, patch_image)
            and float(merge_thr) > 0.0 and int(patch_xy.size(0)) < (1 << 24)
            and (class_thrs is None or isinstance(class_thrs, (list, tuple, torch.Tensor))))


def _tiled_fused(dets, labels, patch_index, patch_xy, patch_image, num_classes, local_thr, merge_thr, class_thrs, group,
                 layout, mark, gather_result):
    """tiled_inference_merge on the fused phases (layouts ``replicated`` and ``routed``)."""
    world, rank = _world(group)
    dev, n, P = dets.device, dets.size(0), int(patch_xy.size(0))
    route_world = world if layout == 'routed' else 1
    if route_world > ROUTE_MAX_WORLD or (world > 1 and (P + 1) * world > RESTORE_MAX_SLOTS):
        return None
    capacity = None
    repeats = 0
    fatal = None
    width = route_world + 2 + _lib.NMS_RESULT_BYTES // 8
    while True:
        # A rank that cannot go on (workspace cap, a phase that keeps asking for a repeat) must not leave the others
        # waiting in the next exchange of the counts: it publishes TILED_ABORT as its K word instead of running its
        # phase, every rank sees it in the gathered matrix, and all of them raise together.
        rows = counts = None
        if fatal is None:
            try:
                rows, counts = tiled_local_phase(dets, labels, patch_index, patch_xy, patch_image, num_classes, local_thr,
                                                 route_world, capacity=capacity)
            except _lib.ObbOpsError as e:
                fatal = e
        if fatal is not None:
            counts = torch.full((width,), TILED_ABORT, dtype=torch.int64, device=dev)
        mark('local')
        if world > 1:
            mat = torch.empty(world * width, dtype=torch.int64, device=dev)
            coll.all_gather_into_tensor(mat, counts, group=group)
            mat = mat.view(world, width).tolist()                       # the exchange's one host read
        else:
            with _lib.on_device(dev):
                mat = [counts.tolist()]
        kwords = [row[route_world + 1] for row in mat]
        if any(k == TILED_ABORT for k in kwords):
            if fatal is not None:
                raise fatal
            raise _lib.ObbOpsError('rank(s) %s aborted the per-patch phase of the tiled pass'
                                   % [r for r, k in enumerate(kwords) if k == TILED_ABORT])
        if all(k >= 0 for k in kwords):
            break
        repeats += 1                          # (counted from the gathered matrix: the same number on every rank)
        if repeats > 4:
            raise _lib.ObbOpsError('the per-patch phase of the tiled pass was repeated %d times (K words %s)' % (repeats - 1, kwords))
        if kwords[rank] == TILED_REPEAT:
            # this rank's NMS gave up on its grid route (a crowded cell): the library notes it for the thread through
            # the record (OBB_ERETRY) and the phase is repeated as it is
            rec = np.asarray(mat[rank][route_world + 2:], dtype=np.int64)
            info = (ctypes.c_int64 * 4)()
            rc = _lib.lib().obb_tiled_result(rec.ctypes.data, 0, n, 6, route_world, -1, int(counts.phase_nbytes), info)
            global phase_repeats
            phase_repeats += 1
            if rc != _lib.OBB_ERETRY:             # (the library's note sends the next try down the general route)
                fatal = _lib.ObbOpsError('the per-patch phase asks to be repeated but its record says rc %d: %s'
                                         % (rc, _lib.lib().obb_last_error().decode()))
            continue
        # a rank's per-patch NMS overflowed its pair lists (every rank sees it): that rank repeats its phase with the
        # capacity its kernels asked for, everybody repeats the exchange of the counts
        if TILED_REPEAT < kwords[rank] < 0:
            capacity = max(-kwords[rank], 2 * (capacity or 0))
            _note_capacity(0, n, capacity)
            nbytes = int(_lib.lib().obb_tiled_local_workspace_bytes(n, route_world, capacity))
            if nbytes > nms_wrapper.MAX_WORKSPACE_BYTES:
                fatal = _capacity_error('the per-patch NMS', n, nbytes, capacity)
    K = kwords[rank]
    if layout == 'routed' and world > 1:
        if any(row[world] for row in mat):
            raise ValueError('order_key: the keys of a rank must be non-decreasing in its row order')
        in_splits, out_splits = mat[rank][:world], [mat[src][rank] for src in range(world)]
        recv = rows.new_empty((sum(out_splits), 9))
        coll.all_to_all_single(recv, rows[:K], out_splits, in_splits, group=group)
        starts = [0]
        for c in out_splits[:-1]:
            starts.append(starts[-1] + c)
        total, key_max = recv.size(0), P
    elif world > 1:                                                       # replicated: everything to everybody
        nmax = max(max(kwords), 1)
        pad = rows.new_zeros((nmax, 9))
        pad[:K] = rows[:K]
        pad[K:, -1] = _PAD_KEY
        recv = rows.new_empty((world * nmax, 9))
        coll.all_gather_into_tensor(recv, pad, group=group)             # cf. test.py:133
        starts, total, key_max = [r * nmax for r in range(world)], sum(kwords), P
    else:
        recv, starts, total, key_max = rows[:K], [0], K, None
    mark('gather')
    want_key = layout == 'routed' and world > 1 and gather_result
    kept, keep, ordered = tiled_merge_phase(recv, starts, key_max, total, num_classes, merge_thr, class_thrs,
                                            cols=8, out_cols=9 if want_key else 8)
    if key_max is not None:
        _check_order_flag(dev)
    allrows = ordered[:, :8]
    if want_key:
        padded, cnts, _ = _gather_padded(kept, group, _PAD_KEY)
        kept = _restore_by_key(padded, sum(cnts), 8, P, world)
    mark('merge')
    return kept, keep, allrows


def tiled_inference_merge(dets, labels, patch_index, seq, patch_xy, patch_image, num_classes,
                          local_thr=0.5, merge_thr=0.3, class_thrs=None, group=None,
                          layout='replicated', nms_fn=None, mark=None, gather_result=True):
    """One pass of BASELINE.json configs[4] over this rank's patches:

        per-patch, per-class rotated NMS of the local detections (one grouped launch)
        -> image coordinates -> all-gather over RCCL -> merged per-(image, class) NMS

    ``mark(name)`` (optional) is called between the phases ('local', 'gather', 'merge'), e.g.
    to record HIP events.  Returns ``distributed_merge``'s triple."""
    mark = mark or (lambda name: None)
    if _fusable(dets, labels, patch_index, patch_xy, patch_image, layout, nms_fn, local_thr, merge_thr, class_thrs):
        res = _tiled_fused(dets, labels, patch_index, patch_xy, patch_image, num_classes, local_thr, merge_thr, class_thrs,
                           group, layout, mark, gather_result)
        if res is not None:
            return res
    keep = local_patch_nms(dets, labels, patch_index, local_thr, num_classes, nms_fn)
    use_patch_key = _world(group)[0] > 1 and patch_index is not None and int(patch_xy.size(0)) < (1 << 24)
    rows = patch_rows_to_image(dets, labels, patch_index, patch_xy, patch_image, keep=keep, with_patch=use_patch_key)
    rows, okey = rows if use_patch_key else (rows, None)
    mark('local')
    # (the order only matters when rows of several ranks are interleaved.  A patch lives on one rank and
    # its rows are contiguous and in order there, so the patch index restores the single-process order;
    # ``seq`` -- arbitrary global row numbers -- is the general, slower way and is used without it)
    skept = None
    if _world(group)[0] > 1 and okey is None and seq is not None:
        skept = seq[keep]
    return distributed_merge(rows, merge_thr, num_classes, class_thrs, group, nms_fn, skept,
                             layout, mark=mark, order_key=okey, order_key_max=int(patch_xy.size(0)),
                             gather_result=gather_result)
