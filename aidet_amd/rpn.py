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
import torch

from . import nms_wrapper
from .bbox_transforms import delta2bbox


def _cfg(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def rpn_proposals(cls_scores, bbox_preds, mlvl_anchors, img_shape, cfg, target_means=(.0, .0, .0, .0),
                  target_stds=(1.0, 1.0, 1.0, 1.0), use_sigmoid_cls=True, nms_op=None):
    """Proposals (n, 5) [x1, y1, x2, y2, score] of one image.

    Args:
        cls_scores (list[Tensor]): per level (A * cls_out, H, W) logits.
        bbox_preds (list[Tensor]): per level (A * 4, H, W) deltas.
        mlvl_anchors (list[Tensor]): per level (H * W * A, 4) anchors (``AnchorGenerator``).
        img_shape (tuple): (H, W, ...) of the network input: boxes are clamped into it.
        cfg: ``nms_pre``, ``nms_post``, ``max_num``, ``nms_thr``, ``min_bbox_size``,
            ``nms_across_levels`` (test_cfg.rpn of configs/dota/dota_v002...:118-124).
    """
    nms_op = nms_op or nms_wrapper.nms
    nms_pre, nms_post, max_num = _cfg(cfg, 'nms_pre'), _cfg(cfg, 'nms_post'), _cfg(cfg, 'max_num')
    boxes, scores, levels = [], [], []
    for lvl, (cls, reg, anchors) in enumerate(zip(cls_scores, bbox_preds, mlvl_anchors)):
        assert cls.size()[-2:] == reg.size()[-2:]
        cls = cls.permute(1, 2, 0)
        if use_sigmoid_cls:
            s = cls.reshape(-1).sigmoid()
        else:
            s = cls.reshape(-1, 2).softmax(dim=1)[:, 1]
        reg = reg.permute(1, 2, 0).reshape(-1, 4)
        if nms_pre > 0 and s.shape[0] > nms_pre:
            s, top = s.topk(nms_pre)
            reg, anchors = reg[top, :], anchors[top, :]
        b = delta2bbox(anchors, reg, target_means, target_stds, img_shape)
        min_size = _cfg(cfg, 'min_bbox_size')
        if min_size > 0:
            ok = ((b[:, 2] - b[:, 0] + 1 >= min_size) & (b[:, 3] - b[:, 1] + 1 >= min_size)).nonzero(as_tuple=True)[0]
            b, s = b[ok, :], s[ok]
        boxes.append(b)
        scores.append(s)
        levels.append(b.new_full((b.size(0), ), lvl, dtype=torch.long))
    dets = torch.cat([torch.cat(boxes), torch.cat(scores)[:, None]], dim=-1)
    levels = torch.cat(levels)
    proposals, keep = nms_op(dets, _cfg(cfg, 'nms_thr'), groups=levels)
    # `proposals[:nms_post]` per level: rank of every survivor inside its level
    lv = levels[keep]
    first = torch.searchsorted(lv, torch.arange(len(cls_scores), device=lv.device))
    rank = torch.arange(lv.numel(), device=lv.device) - first[lv]
    proposals = proposals[rank < nms_post]
    if _cfg(cfg, 'nms_across_levels'):
        proposals, _ = nms_op(proposals, _cfg(cfg, 'nms_thr'))
        return proposals[:max_num, :]
    num = min(max_num, proposals.shape[0])
    _, top = proposals[:, 4].topk(num)
    return proposals[top, :]
