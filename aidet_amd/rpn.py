"""RPN proposal generation for one image (host orchestration + the HBB NMS kernel).

Counterpart of ``RPNHead.get_bboxes_single`` (mmdet/models/anchor_heads/rpn_head.py:55-104):
per FPN level -- sigmoid, top ``nms_pre`` by score, ``delta2bbox`` into the image, NMS at
``nms_thr``, first ``nms_post`` -- then the best ``max_num`` of all levels.

The reference runs five NMS calls, each with a device-to-host copy of the suppression mask
and a host loop (nms_kernel.cu:104-131).  Here the levels are concatenated and decided by ONE
grouped launch (group = level): boxes of different levels never interact, and because the
keep list comes back ascending in the concatenated index, it IS the reference's
``torch.cat(mlvl_proposals)`` order (level-major, descending score inside a level).
"""
import ctypes

import torch

from . import _lib, nms_wrapper
from .bbox_transforms import delta2bbox

SEL_MAX = 2048     # obb_rpn_select_decode_f32 / obb_rows_topk_f32 keep at most this many rows per selection


def _cfg(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


FUSED = True    # False: the tensor-op path below also on the GPU (the specification the fused path is tested against)


def _rpn_proposals_fused(cls_scores, bbox_preds, mlvl_anchors, img_shape, nms_pre, nms_post, max_num, nms_thr,
                         means, stds, use_sigmoid):
    """The same proposals from four launches + one NMS call: ``obb_rpn_select_decode_ws_f32`` (per level: exact top
    ``nms_pre`` by score, delta2bbox, clamp), the grouped horizontal NMS, ``obb_rows_topk_f32`` (per-level cut at
    ``nms_post``, top ``max_num`` by score)."""
    L = _lib.lib()
    dev = cls_scores[0].device
    n = len(cls_scores)
    cls = [c.detach().contiguous() for c in cls_scores]
    reg = [r.detach().contiguous() for r in bbox_preds]
    anc = [a.detach().contiguous() for a in mlvl_anchors]
    A = reg[0].size(0) // 4
    hs = (ctypes.c_int * n)(*[c.size(-2) for c in cls])
    wd = (ctypes.c_int * n)(*[c.size(-1) for c in cls])
    per_level = [min(nms_pre, c.size(-2) * c.size(-1) * A) for c in cls]
    N = sum(per_level)
    if N == 0:
        return cls[0].new_zeros((0, 5))
    dets = torch.empty((N, 5), dtype=torch.float32, device=dev)
    levels = torch.empty(N, dtype=torch.int32, device=dev)
    rows = ctypes.c_int64(0)
    pc = (ctypes.c_void_p * n)(*[c.data_ptr() for c in cls])
    pr = (ctypes.c_void_p * n)(*[r.data_ptr() for r in reg])
    pa = (ctypes.c_void_p * n)(*[a.data_ptr() for a in anc])
    mh, mw = (int(img_shape[0]), int(img_shape[1])) if img_shape is not None else (0, 0)
    with _lib.on_device(dev):
        st = _lib.current_stream_handle(dev)
        wsb = int(L.obb_rpn_select_workspace_bytes(hs, wd, n, A, int(nms_pre)))    # (> 0: large levels, several workgroups each)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        _lib.check(L.obb_rpn_select_decode_ws_f32(pc, pr, pa, hs, wd, n, A, 1 if use_sigmoid else 0, int(nms_pre),
                                                  (ctypes.c_float * 4)(*[float(v) for v in means]),
                                                  (ctypes.c_float * 4)(*[float(v) for v in stds]), 16 / 1000.0, mh, mw,
                                                  dets.data_ptr(), levels.data_ptr(), ctypes.byref(rows),
                                                  ws.data_ptr() if ws is not None else None, wsb, st))
        assert rows.value == N
        keep, _ = nms_wrapper._run_nms(dets, _lib.KIND_HBB, float(nms_thr), levels)
        K = int(keep.numel())
        base = [0]
        for m in per_level:
            base.append(base[-1] + m)
        out = torch.empty((max_num, 5), dtype=torch.float32, device=dev)
        count = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(L.obb_rows_topk_f32(dets.data_ptr(), N, 5, 4, keep.data_ptr(), K, (ctypes.c_int * (n + 1))(*base), n,
                                       int(nms_post), int(max_num), out.data_ptr(), None, count.data_ptr(), st))
    # every level keeps all its survivors when it has at most nms_post candidates: the count is then known here
    num = min(max_num, K) if all(m <= nms_post for m in per_level) else int(count.item())
    return out[:num]


def rpn_proposals(cls_scores, bbox_preds, mlvl_anchors, img_shape, cfg, target_means=(.0, .0, .0, .0),
                  target_stds=(1.0, 1.0, 1.0, 1.0), use_sigmoid_cls=True, nms_op=None, fused=None):
    """Proposals (n, 5) [x1, y1, x2, y2, score] of one image.

    Args:
        cls_scores (list[Tensor]): per level (A * cls_out, H, W) logits.
        bbox_preds (list[Tensor]): per level (A * 4, H, W) deltas.
        mlvl_anchors (list[Tensor]): per level (H * W * A, 4) anchors (``AnchorGenerator``).
        img_shape (tuple): (H, W, ...) of the network input: boxes are clamped into it.
        cfg: ``nms_pre``, ``nms_post``, ``max_num``, ``nms_thr``, ``min_bbox_size``,
            ``nms_across_levels`` (test_cfg.rpn of configs/dota/dota_v002...:118-124).
        fused (bool, optional): the HIP selection kernels (``_rpn_proposals_fused``) where they apply (fp32 HIP
            tensors, ``nms_pre`` / ``max_num`` <= 2048, no ``min_bbox_size``, no ``nms_across_levels``); default: the
            module switch ``FUSED``.  False: the tensor-op path below (the specification).
    """
    nms_pre, nms_post, max_num = _cfg(cfg, 'nms_pre'), _cfg(cfg, 'nms_post'), _cfg(cfg, 'max_num')
    min_size = _cfg(cfg, 'min_bbox_size')
    num_levels = len(cls_scores)
    if (nms_op is None and (FUSED if fused is None else fused) and min_size <= 0 and not _cfg(cfg, 'nms_across_levels') and 0 < nms_pre <= SEL_MAX
            and 0 < max_num <= SEL_MAX and num_levels <= 8
            and all(t.is_cuda and t.dtype == torch.float32 for t in list(cls_scores) + list(bbox_preds) + list(mlvl_anchors))):
        return _rpn_proposals_fused(cls_scores, bbox_preds, mlvl_anchors, img_shape, nms_pre, nms_post, max_num,
                                    _cfg(cfg, 'nms_thr'), target_means, target_stds, use_sigmoid_cls)
    nms_op = nms_op or nms_wrapper.nms
    lvl_scores, lvl_regs = [], []
    for cls, reg in zip(cls_scores, bbox_preds):
        assert cls.size()[-2:] == reg.size()[-2:]
        cls = cls.permute(1, 2, 0)
        if use_sigmoid_cls:
            lvl_scores.append(cls.reshape(-1).sigmoid())
        else:
            lvl_scores.append(cls.reshape(-1, 2).softmax(dim=1)[:, 1])
        lvl_regs.append(reg.permute(1, 2, 0).reshape(-1, 4))
    # Levels with more than nms_pre anchors keep their nms_pre best (in descending score order,
    # rpn_head.py:77-81) -- ONE batched top-k over those levels padded to a common length, not a
    # top-k launch per level; smaller levels pass through in anchor order.
    big = [i for i, s in enumerate(lvl_scores) if nms_pre > 0 and s.shape[0] > nms_pre]
    picked = [None] * num_levels
    if big:
        width = max(lvl_scores[i].shape[0] for i in big)
        padded = torch.stack([torch.nn.functional.pad(lvl_scores[i], (0, width - lvl_scores[i].shape[0]),
                                                      value=float('-inf')) for i in big])
        top_s, top_i = padded.topk(nms_pre, dim=1)
        for row, i in enumerate(big):
            picked[i] = (top_s[row], lvl_regs[i][top_i[row]], mlvl_anchors[i][top_i[row]])
    for i in range(num_levels):
        if picked[i] is None:
            picked[i] = (lvl_scores[i], lvl_regs[i], mlvl_anchors[i])
    scores = torch.cat([p[0] for p in picked])
    levels = torch.cat([p[0].new_full((p[0].shape[0], ), i, dtype=torch.long) for i, p in enumerate(picked)])
    # one decode for all levels (same means / stds / image clamp everywhere)
    boxes = delta2bbox(torch.cat([p[2] for p in picked]), torch.cat([p[1] for p in picked]),
                       target_means, target_stds, img_shape)
    if min_size > 0:
        ok = ((boxes[:, 2] - boxes[:, 0] + 1 >= min_size) & (boxes[:, 3] - boxes[:, 1] + 1 >= min_size)).nonzero(as_tuple=True)[0]
        boxes, scores, levels = boxes[ok, :], scores[ok], levels[ok]
    dets = torch.cat([boxes, scores[:, None]], dim=-1)
    proposals, keep = nms_op(dets, _cfg(cfg, 'nms_thr'), groups=levels)
    # `proposals[:nms_post]` per level: rank of every survivor inside its level
    lv = levels[keep]
    first = torch.searchsorted(lv, torch.arange(num_levels, device=lv.device))
    rank = torch.arange(lv.numel(), device=lv.device) - first[lv]
    proposals = proposals[rank < nms_post]
    if _cfg(cfg, 'nms_across_levels'):
        proposals, _ = nms_op(proposals, _cfg(cfg, 'nms_thr'))
        return proposals[:max_num, :]
    num = min(max_num, proposals.shape[0])
    _, top = proposals[:, 4].topk(num)
    return proposals[top, :]
