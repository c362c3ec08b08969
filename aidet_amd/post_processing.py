"""Class-wise NMS orchestration for horizontal and rotated detections.

Counterparts of the reference's mmdet/core/post_processing/bbox_nms.py
(``multiclass_nms`` :6-76) and rbbox_nms.py (``multiclass_nms_with_index`` :6-62,
``thetaobb_nms_by_bbox_nms`` :64-119), with the NMS op resolved by name exactly as
the reference does (``getattr(nms_wrapper, nms_cfg['type'])``, bbox_nms.py:60-62).

Design: every function first *flattens* the (box, class) candidates that pass the
score threshold into one list and hands that list to ONE grouped NMS launch
(group = class).  The reference's per-class Python loop (15 launches, each with a
device-to-host mask copy, rbbox_nms.py:29-49) disappears; what the loop returned --
per-class keep lists, class-major concatenation -- is recovered from the single
ascending keep list by a segmented split.

``multiclass_rnms`` is the call the reference left commented out
(``multiclass_thetaobb_nms(thetaobbs, scores, cfg.score_thr,
cfg.polygon_nms_iou_thr, cfg.max_per_img)``, rbbox_head.py:294-295).
"""
import ctypes

import numpy as np
import torch

from . import _lib, nms_wrapper


def _per_class_boxes(multi_boxes, num_fg, d):
    """(n, (1+num_fg)*d) | (n, d)  ->  (n, num_fg, d) view of the foreground classes."""
    n = multi_boxes.size(0)
    if multi_boxes.size(1) > d:
        return multi_boxes.view(n, -1, d)[:, 1:]
    return multi_boxes[:, None].expand(n, num_fg, d)


def _resolve_op(nms_cfg, default):
    cfg = dict(nms_cfg)
    return getattr(nms_wrapper, cfg.pop('type', default)), cfg


def _top_by_score(dets, labels, max_num):
    """The reference's cap (bbox_nms.py:68-74): only when more than ``max_num`` rows survive are
    they re-ordered by descending score and cut.  (``max_num=-1`` therefore drops the
    lowest-scored row, as ``inds[:-1]`` does in the reference.)"""
    if dets.size(0) > max_num:
        order = dets[:, -1].sort(descending=True)[1][:max_num]
        dets, labels = dets[order], labels[order]
    return dets, labels


def _multiclass(multi_boxes, multi_scores, score_thr, nms_cfg, max_num, score_factors, d,
                default_op, offset_trick, report=None):
    """``report`` = (multi_boxes_out, d_out): the NMS decides on ``multi_boxes`` but the returned rows are the
    boxes of the same (row, class) in ``multi_boxes_out`` (another encoding of the same detections)."""
    num_fg = multi_scores.size(1) - 1
    boxes = _per_class_boxes(multi_boxes, num_fg, d)
    scores = multi_scores[:, 1:]
    valid = scores > score_thr
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    rows, labels = valid.nonzero(as_tuple=True)          # row-major, like bbox_nms.py:43
    shown = boxes if report is None else _per_class_boxes(report[0], num_fg, report[1])
    boxes, scores, shown = boxes[rows, labels], scores[rows, labels], shown[rows, labels]
    if boxes.numel() == 0:
        return multi_boxes.new_zeros((0, shown.size(-1) + 1)), multi_boxes.new_zeros((0, ), dtype=torch.long)
    nms_op, cfg = _resolve_op(nms_cfg, default_op)
    if offset_trick:
        # bbox_nms.py:56-59 verbatim in effect: the classes are pushed apart by
        # label * (max coordinate + 1), so the op sees the same floats as the reference's
        shift = labels.to(boxes) * (boxes.max() + 1)
        dets, keep = nms_op(torch.cat([boxes + shift[:, None], scores[:, None]], 1), **cfg)
    elif nms_op is nms_wrapper.soft_nms:
        # grouped Soft-NMS = every class an array of its own (soft_nms_grouped); the candidates are put into class-major
        # order for it and `keep` is mapped back to the row-major list
        if d != 4:
            raise ValueError("type='soft_nms' is a horizontal-box op (nms_cpu.cpp:71-210): it cannot decide rotated boxes")
        order = torch.argsort(labels, stable=True)
        sizes = torch.bincount(labels, minlength=num_fg)
        res = nms_wrapper.soft_nms_grouped(torch.cat([boxes, scores[:, None]], 1)[order], sizes, **cfg)
        starts = (torch.cumsum(sizes, 0) - sizes).tolist()
        dets = torch.cat([r[0] for r in res]) if res else boxes.new_zeros((0, 5))
        keep = torch.cat([order[r[1] + st] for r, st in zip(res, starts)]) if res else order[:0]
    else:
        # same per-class NMS without moving (and thereby rounding) a coordinate
        dets, keep = nms_op(torch.cat([boxes, scores[:, None]], 1), groups=labels, **cfg)
    out = torch.cat([shown[keep], dets[:, -1:]], 1)      # dets[:, -1]: soft-NMS rescored
    labels = labels[keep]
    if max_num > 0 or offset_trick:   # (the new rotated op does not inherit the max_num=-1 quirk)
        out, labels = _top_by_score(out, labels, max_num)
    return out, labels


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None,
                   grouped=False):
    """Horizontal multi-class NMS, ``multiclass_nms`` of bbox_nms.py:6-76.

    Args:
        multi_bboxes (Tensor): (n, #class*4) or (n, 4)
        multi_scores (Tensor): (n, #class); column 0 is the background and ignored
        score_thr (float): candidates need ``score > score_thr``
        nms_cfg (dict): e.g. ``dict(type='nms', iou_thr=0.5)``
        max_num (int): if more survive, the ``max_num`` best by score are returned
        score_factors (Tensor, optional): multiplied to the scores before NMS
        grouped (bool): False (default) separates classes with the reference's coordinate
            offsets (:56-59), i.e. the NMS sees exactly the reference's floats; True passes
            the class as group id instead (no coordinate is rounded by an offset).

    Returns:
        tuple: (bboxes (k, 5), labels (k,)); labels are 0-based.
    """
    return _multiclass(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num, score_factors, 4,
                       'nms', offset_trick=not grouped)


def multiclass_rnms(multi_rbboxes, multi_scores, score_thr, nms_cfg, max_num=-1,
                    score_factors=None, out_dim_reg=5, report=None):
    """Rotated NMS for multi-class theta-OBBs: the contract of ``multiclass_nms``
    (bbox_nms.py:6-76) with ``out_dim_reg`` numbers per box and ``type='rnms'``.

    Returns:
        tuple: (rbboxes (k, out_dim_reg + 1), labels (k,)); labels are 0-based.
    """
    return _multiclass(multi_rbboxes, multi_scores, score_thr, nms_cfg, max_num, score_factors,
                       out_dim_reg, 'rnms', offset_trick=False, report=report)


def multiclass_thetaobb_nms(thetaobbs, scores, score_thr, polygon_nms_iou_thr, max_per_img=-1):
    """The call the reference left commented out in ``RBBoxHead.get_det_rbboxes_parallel``
    (mmdet/models/bbox_heads/rbbox_head.py:294-295), with ITS positional signature:
    ``multiclass_thetaobb_nms(thetaobbs, scores, cfg.score_thr, cfg.polygon_nms_iou_thr, cfg.max_per_img)`` ->
    ``(det_thetaobbs (k, 6), det_labels (k,))`` = ``multiclass_rnms`` with ``dict(type='rnms', iou_thr=...)``."""
    return multiclass_rnms(thetaobbs, scores, score_thr, dict(type='rnms', iou_thr=polygon_nms_iou_thr), max_per_img)


def _class_major(masks):
    """masks (C, n) bool -> (class, row) of every set entry, class-major; counts per class."""
    cls, rows = masks.nonzero(as_tuple=True)
    return cls, rows, masks.sum(1)


def multiclass_nms_with_index(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """Horizontal multi-class NMS that also reports what it did per class
    (rbbox_nms.py:6-62), so that the decision can be re-applied to the rotated boxes.

    One grouped launch over the class-major list of candidates replaces the loop over
    classes.  The keep list of that launch is ascending in the class-major index, which
    is the order in which the reference concatenates its per-class results; the per-class
    keep lists are its segments, re-based to the start of their class.

    Returns:
        tuple: (bboxes (k, 5), labels (k,), bbox_cls_inds, bbox_keep_inds) --
        ``bbox_cls_inds``: one (n,) bool mask per foreground class (``score > score_thr``);
        ``bbox_keep_inds``: for every class with at least one candidate, in class order,
        the int64 indices the NMS kept among that class's candidates.
    """
    num_fg = multi_scores.size(1) - 1
    masks = (multi_scores[:, 1:] > score_thr).t().contiguous()          # (C, n)
    bbox_cls_inds = list(masks.unbind(0))
    cls, rows, per_class = _class_major(masks)
    if rows.numel() == 0:
        return (multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long),
                bbox_cls_inds, [])
    boxes = _per_class_boxes(multi_bboxes, num_fg, 4)[rows, cls]
    dets = torch.cat([boxes, multi_scores[rows, cls + 1][:, None]], 1)
    nms_op, cfg = _resolve_op(nms_cfg, 'nms')
    if nms_op is nms_wrapper.soft_nms:
        # Soft-NMS re-scores and re-orders its input: class by class, as the reference does for every op
        # (rbbox_nms.py:29-49) -- one workgroup per class in ONE launch on the GPU (the classes are independent
        # arrays there too), the Python loop with one call per class elsewhere
        sizes = per_class.tolist()
        present = [c for c, m in enumerate(sizes) if m]
        if dets.is_cuda:
            res = nms_wrapper.soft_nms_grouped(dets, [sizes[c] for c in present], **cfg)
        else:
            res, start = [], 0
            for c in present:
                res.append(nms_op(dets[start:start + sizes[c]], **cfg))
                start += sizes[c]
        kept = torch.cat([d for d, _ in res])
        bbox_keep_inds = [k for _, k in res]
        labels = torch.cat([cls.new_full((len(k), ), c) for c, k in zip(present, bbox_keep_inds)])
    else:
        kept, keep = nms_op(dets, groups=cls, **cfg)
        labels = cls[keep]
        # segments of the keep list: candidates and survivors per class in ONE host read
        counts = torch.stack([per_class, torch.bincount(labels, minlength=num_fg)]).tolist()
        bbox_keep_inds, start, kstart = [], 0, 0
        for m, k in zip(*counts):
            if m:
                bbox_keep_inds.append(keep[kstart:kstart + k] - start)
            start += m
            kstart += k
    kept, labels = _top_by_score(kept, labels, max_num)
    return kept, labels, bbox_cls_inds, bbox_keep_inds


def thetaobb_nms_by_bbox_nms(multi_bboxes, multi_scores, bbox_cls_inds, bbox_keep_inds,
                             max_num=-1, out_dim_reg=5):
    """The reference's *stand-in* for rotated NMS (rbbox_nms.py:64-119): the rotated
    boxes that survive are those whose horizontal boxes survived.  Here a single gather:
    the per-class keep lists of ``multiclass_nms_with_index`` are re-based into the
    class-major candidate list and select (row, class) pairs directly.

    Like the reference (:99, ``pop(0)``) the keep lists that were used are removed from
    ``bbox_keep_inds``.
    """
    d = out_dim_reg
    num_fg = multi_scores.size(1) - 1
    empty = (multi_bboxes.new_zeros((0, d + 1)), multi_bboxes.new_zeros((0, ), dtype=torch.long))
    if num_fg == 0:
        return empty
    masks = torch.stack(list(bbox_cls_inds[:num_fg]))                   # (C, n)
    cls, rows, per_class = _class_major(masks)
    parts, start, used = [], 0, 0
    for m in per_class.tolist():
        if m:
            parts.append(bbox_keep_inds[used] + start)
            used += 1
        start += m
    del bbox_keep_inds[:used]
    if not parts:
        return empty
    sel = torch.cat(parts)
    rows, cls = rows[sel], cls[sel]
    boxes = _per_class_boxes(multi_bboxes, num_fg, d)[rows, cls]
    dets = torch.cat([boxes, multi_scores[rows, cls + 1][:, None]], 1)
    return _top_by_score(dets, cls, max_num)


def get_det_rbboxes(rois, cls_score, rbbox_pred, img_shape, scale_factor, rescale=False, cfg=None,
                    target_means=(0., 0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2, 0.1),
                    bbox_cls_inds=None, bbox_keep_inds=None, rotated_nms=True, encode=None):
    """Oriented detections of one image from the rbbox head's outputs.

    Counterpart of ``RBBoxHead.get_det_rbboxes_parallel``
    (mmdet/models/bbox_heads/rbbox_head.py:253-296): softmax, the decoder of the head's
    encoding with the head's means/stds, optional rescale, then NMS.  ``encode`` (default:
    ``cfg.encode`` if the config has one, else 'thetaobb', the encoding of config 4) selects
    ``delta2thetaobb`` / ``delta2pointobb`` / ``delta2hobb`` and the matching rescale, as the
    reference's two dictionaries do (:264-269); ``target_means`` / ``target_stds`` must have the
    encoding's length (5 / 8 / 5).

    ``rotated_nms=True`` runs the call the reference left commented out (:294-295):
    ``multiclass_rnms(thetaobbs, scores, cfg.score_thr, polygon_nms_iou_thr,
    max_per_img)``.  ``rotated_nms=False`` reproduces the reference's behaviour, i.e.
    re-applies the horizontal keep indices (:287-293) -- then ``bbox_cls_inds`` /
    ``bbox_keep_inds`` from ``multiclass_nms_with_index`` are required.  The true NMS of the
    'pointobb' encoding is ``poly_nms`` on the decoded corner points; 'hobb' boxes are suppressed as the
    quadrilaterals ``rbbox_transforms.hobb_corner_points`` builds from them (the returned rows stay hobbs).

    ``cfg`` needs ``score_thr``, ``max_per_img`` and ``polygon_nms_iou_thr`` (attribute or
    key access), as test_cfg.rcnn of configs/dota/dota_v002_theta_obb_r50_v1_train.py:125-132.
    """
    import torch.nn.functional as F

    from . import rbbox_transforms as T

    def get(name):
        return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)

    if encode is None:
        encode = 'thetaobb'
        if cfg is not None and (('encode' in cfg) if isinstance(cfg, dict) else hasattr(cfg, 'encode')):
            encode = get('encode')
    formats = {'thetaobb': (T.delta2thetaobb, T.thetaobb_rescale, 5, 'rnms'),
               'pointobb': (T.delta2pointobb, T.pointobb_rescale, 8, 'poly_nms'),
               'hobb': (T.delta2hobb, T.hobb_rescale, 5, 'poly_nms')}
    if encode not in formats:
        raise ValueError('encode must be one of {}, got {!r}'.format(sorted(formats), encode))
    decode, rescale_fn, dim, rotated_op = formats[encode]

    if isinstance(cls_score, list):
        cls_score = sum(cls_score) / float(len(cls_score))
    scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
    if rbbox_pred is not None:
        rbboxes = decode(rois[:, 1:], rbbox_pred, target_means, target_stds, img_shape)
    else:
        rbboxes = rois[:, 1:]
    if rescale:
        rbboxes = rescale_fn(rbboxes, scale_factor, reverse_flag=True)
    if cfg is None:
        return rbboxes, scores
    if rotated_nms:
        nms_cfg = dict(type=rotated_op, iou_thr=get('polygon_nms_iou_thr'))
        if encode != 'hobb':
            return multiclass_rnms(rbboxes, scores, get('score_thr'), nms_cfg, get('max_per_img'), out_dim_reg=dim)
        # hobb: the NMS sees each box's quadrilateral, the rows that come back are the hobbs themselves
        n, k = rbboxes.size(0), rbboxes.size(1) // 5
        quads = T.hobb_corner_points(rbboxes.view(n, k, 5)).reshape(n, k * 8)
        return multiclass_rnms(quads, scores, get('score_thr'), nms_cfg, get('max_per_img'), out_dim_reg=8,
                               report=(rbboxes, 5))
    return thetaobb_nms_by_bbox_nms(rbboxes, scores, bbox_cls_inds, bbox_keep_inds,
                                    get('max_per_img'), out_dim_reg=dim)


# ---- the same stages as fused launches (the detector's inference path) -------------------------------------------------
HEAD_MAX_ROIS, HEAD_MAX_FG, TOPK_MAX = 2048, 16, 2048      # capacities of obb_head_select_decode_f32 / obb_rows_topk_f32


def head_fusable(rois, cls_score, pred, max_num, scale_factor=None):
    """True when the fused head kernels serve this call: fp32 HIP tensors, at most 2048 RoIs / 16 foreground classes,
    a cap of at most 2048 detections, a SCALAR scale factor (a per-axis one -- the 4-element array of
    ``keep_ratio=False``, bbox_head.py:160-166 -- goes through the tensor-op stages)."""
    ts = (rois, cls_score, pred)
    if scale_factor is not None and not isinstance(scale_factor, (int, float)) and np.ndim(scale_factor) != 0:
        return False
    return (all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in ts)
            and rois.size(0) <= HEAD_MAX_ROIS and 2 <= cls_score.size(1) <= HEAD_MAX_FG + 1 and 0 < max_num <= TOPK_MAX
            and not torch.is_grad_enabled())


def _codec(dim, means, stds, img_shape, scale_factor):
    m = (ctypes.c_float * 5)(*([float(v) for v in means] + [0.0] * (5 - dim)))
    sd = (ctypes.c_float * 5)(*([float(v) for v in stds] + [1.0] * (5 - dim)))
    mh, mw = (int(img_shape[0]), int(img_shape[1])) if (img_shape is not None and dim == 4) else (0, 0)
    return m, sd, mh, mw, float(scale_factor) if (scale_factor is not None and float(scale_factor) != 0.0) else 0.0


def select_decode_candidates(rois, cls_score, pred, dim, score_thr, means, stds, img_shape=None, scale_factor=None):
    """softmax -> ``score > score_thr`` -> decode of the listed (RoI, class) pairs, ONE launch
    (``obb_head_select_decode_f32``): what ``get_det_bboxes`` / ``get_det_rbboxes_parallel`` compute before their NMS
    and what ``multiclass_nms_with_index`` lists, without decoding every class of every RoI.

    Returns (dets (M, dim + 1), rows int32 (M,), labels int32 (M,), per-class counts (list)), class-major."""
    L = _lib.lib()
    dev = rois.device
    r, c, p = rois.detach().contiguous(), cls_score.detach().contiguous(), pred.detach().contiguous()
    n, C1 = c.size(0), c.size(1)
    cap = max(n * (C1 - 1), 1)
    dets = torch.empty((cap, dim + 1), dtype=torch.float32, device=dev)
    rows = torch.empty(cap, dtype=torch.int32, device=dev)
    labels = torch.empty(cap, dtype=torch.int32, device=dev)
    counts = torch.empty(C1, dtype=torch.int64, device=dev)
    m, sd, mh, mw, sf = _codec(dim, means, stds, img_shape, scale_factor)
    with _lib.on_device(dev):
        _lib.check(L.obb_head_select_decode_f32(r.data_ptr(), c.data_ptr(), p.data_ptr(), n, C1, p.size(1), dim, float(score_thr),
                                                m, sd, 16 / 1000.0, mh, mw, sf, dets.data_ptr(), rows.data_ptr(),
                                                labels.data_ptr(), counts.data_ptr(), _lib.current_stream_handle(dev)))
    counts = counts.tolist()             # (the one host read of the stage: sizes the NMS call)
    M = counts[-1]
    return dets[:M], rows[:M], labels[:M], counts[:-1]


def gather_decode(rois, cls_score, pred, dim, rows, labels, sel, means, stds, img_shape=None, scale_factor=None):
    """Decoded box + softmax score of another head for the (RoI, class) pairs ``rows[sel]``, ``labels[sel]``
    (``obb_head_gather_decode_f32``) -> (K, dim + 1)."""
    L = _lib.lib()
    dev = rois.device
    r, c, p = rois.detach().contiguous(), cls_score.detach().contiguous(), pred.detach().contiguous()
    K = int(sel.numel()) if sel is not None else int(rows.numel())
    out = torch.empty((K, dim + 1), dtype=torch.float32, device=dev)
    m, sd, mh, mw, sf = _codec(dim, means, stds, img_shape, scale_factor)
    with _lib.on_device(dev):
        _lib.check(L.obb_head_gather_decode_f32(r.data_ptr(), c.data_ptr(), p.data_ptr(), c.size(1), p.size(1), dim,
                                                rows.data_ptr(), labels.data_ptr(), None if sel is None else sel.data_ptr(), K,
                                                m, sd, 16 / 1000.0, mh, mw, sf, out.data_ptr(), _lib.current_stream_handle(dev)))
    return out


def top_rows_by_score(rows, keep, max_num):
    """``rows[keep]``, cut to the ``max_num`` best by the last column when more survive (then in descending score order,
    bbox_nms.py:68-74) -> (rows_out, source indices into ``rows``); ``keep`` = None: all rows."""
    dev = rows.device
    if keep is None:
        keep = torch.arange(rows.size(0), device=dev)
    K = int(keep.numel())
    if K <= max_num:
        return _take(rows, keep), keep
    L = _lib.lib()
    out = torch.empty((max_num, rows.size(1)), dtype=torch.float32, device=dev)
    idx = torch.empty(max_num, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    with _lib.on_device(dev):
        _lib.check(L.obb_rows_topk_f32(rows.data_ptr(), rows.size(0), rows.size(1), rows.size(1) - 1, keep.data_ptr(), K, None, 0, 0,
                                       int(max_num), out.data_ptr(), idx.data_ptr(), count.data_ptr(), _lib.current_stream_handle(dev)))
    return out, idx             # (K > max_num valid rows: exactly max_num come back)


def _take(rows, idx):
    return rows[idx] if idx.numel() else rows[:0]


def detect_horizontal_fused(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale, cfg, target_means, target_stds):
    """``get_det_bboxes`` + ``multiclass_nms_with_index`` of one image in four launches: candidates (select + decode),
    grouped horizontal NMS, top-``max_per_img``.  Returns (det_bboxes (k, 5), det_labels (k,), selection) where
    ``selection`` = (rows, labels, keep) names the kept (RoI, class) pairs for the rotated head's keep re-use."""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    nms_cfg = dict(get('nms'))
    op = nms_cfg.pop('type', 'nms')
    assert op == 'nms', 'the fused head path serves the hard horizontal NMS'
    dets, rows, labels, _ = select_decode_candidates(rois, cls_score, bbox_pred, 4, get('score_thr'), target_means, target_stds,
                                                     img_shape, scale_factor if rescale else None)
    if dets.size(0) == 0:
        e = torch.zeros(0, dtype=torch.long, device=rois.device)
        return dets.new_zeros((0, 5)), e, (rows, labels, e)
    keep, _ = nms_wrapper._run_nms(dets, _lib.KIND_HBB, float(nms_cfg['iou_thr']), labels)
    out, idx = top_rows_by_score(dets, keep, get('max_per_img'))
    return out, labels[idx].long(), (rows, labels, keep)


def detect_rotated_fused(rois, cls_score, rbbox_pred, img_shape, scale_factor, rescale, cfg, target_means, target_stds,
                         selection=None):
    """``get_det_rbboxes`` of one image, theta-OBB encoding: ``selection`` given -> the reference's keep re-use (the
    rotated head's own score and decoded theta-OBB of every kept (RoI, class) pair, rbbox_nms.py:64-119); else the true
    rotated NMS (``multiclass_rnms``): candidates, grouped ``rnms``, top-``max_per_img``."""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    sf = scale_factor if rescale else None
    if selection is not None:
        rows, labels, keep = selection
        if keep.numel() == 0:
            return rois.new_zeros((0, 6)), keep
        dets = gather_decode(rois, cls_score, rbbox_pred, 5, rows, labels, keep, target_means, target_stds, img_shape, sf)
        out, idx = top_rows_by_score(dets, None, get('max_per_img'))
        return out, labels[keep[idx]].long()
    dets, rows, labels, _ = select_decode_candidates(rois, cls_score, rbbox_pred, 5, get('score_thr'), target_means, target_stds,
                                                     img_shape, sf)
    if dets.size(0) == 0:
        return dets.new_zeros((0, 6)), torch.zeros(0, dtype=torch.long, device=rois.device)
    keep, _ = nms_wrapper._run_nms(dets, _lib.KIND_RBOX, float(get('polygon_nms_iou_thr')), labels)
    out, idx = top_rows_by_score(dets, keep, get('max_per_img'))
    return out, labels[idx].long()
