"""Class-wise NMS orchestration for rotated detections.

Counterparts of mmdet/core/post_processing/bbox_nms.py (multiclass_nms, :6-76) and
rbbox_nms.py (multiclass_nms_with_index :6-62, thetaobb_nms_by_bbox_nms :64-119)
of the reference, with the NMS op resolved by name exactly as the reference does
(``getattr(nms_wrapper, nms_cfg['type'])``, bbox_nms.py:60-62).

``multiclass_rnms`` is the call the reference left commented out
(``multiclass_thetaobb_nms(thetaobbs, scores, cfg.score_thr,
cfg.polygon_nms_iou_thr, cfg.max_per_img)``, rbbox_head.py:294-295): true
rotated NMS per class, done in ONE grouped kernel call (group = class) instead
of 15 launches with a host sync each (rbbox_nms.py:29-49).
"""
import torch

from . import nms_wrapper


def multiclass_rnms(multi_rbboxes, multi_scores, score_thr, nms_cfg, max_num=-1,
                    score_factors=None, out_dim_reg=5):
    """Rotated NMS for multi-class theta-OBBs.

    Same contract as ``multiclass_nms`` (bbox_nms.py:6-76) with 5 numbers per box.

    Args:
        multi_rbboxes (Tensor): (n, #class*5) or (n, 5)
        multi_scores (Tensor): (n, #class); column 0 is background and ignored
        score_thr (float): boxes with a score <= score_thr are dropped
        nms_cfg (dict): e.g. dict(type='rnms', iou_thr=0.5)
        max_num (int): keep at most max_num detections (highest scores)
        score_factors (Tensor, optional): multiplied to the scores before NMS

    Returns:
        tuple: (rbboxes (k, 6), labels (k,)); labels are 0-based.
    """
    d = out_dim_reg
    num_classes = multi_scores.size(1) - 1
    if multi_rbboxes.shape[1] > d:
        rb = multi_rbboxes.view(multi_scores.size(0), -1, d)[:, 1:]
    else:
        rb = multi_rbboxes[:, None].expand(-1, num_classes, d)
    scores = multi_scores[:, 1:]
    valid_mask = scores > score_thr
    rb = rb[valid_mask]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid_mask]
    labels = valid_mask.nonzero()[:, 1]

    if rb.numel() == 0:
        return (multi_rbboxes.new_zeros((0, d + 1)),
                multi_rbboxes.new_zeros((0, ), dtype=torch.long))

    nms_cfg_ = nms_cfg.copy()
    nms_type = nms_cfg_.pop('type', 'rnms')
    nms_op = getattr(nms_wrapper, nms_type)
    # bbox_nms.py:56-59 separates classes by adding label*(max_coordinate+1) to the
    # coordinates; the grouped kernel gives the same per-class NMS without moving
    # (and thereby rounding) any coordinate.
    dets, keep = nms_op(torch.cat([rb, scores[:, None]], 1), groups=labels, **nms_cfg_)
    rb = rb[keep]
    scores = dets[:, -1]
    labels = labels[keep]
    if keep.size(0) > max_num > 0:
        _, inds = scores.sort(descending=True)
        inds = inds[:max_num]
        rb, scores, labels = rb[inds], scores[inds], labels[inds]
    return torch.cat([rb, scores[:, None]], 1), labels


def multiclass_nms_with_index(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """The reference's horizontal multi-class NMS that also returns, per class, the
    score mask and the keep indices (rbbox_nms.py:6-62).  Per-class Python loop and
    concatenation order are kept so that outputs are identical."""
    num_classes = multi_scores.shape[1]
    bboxes, labels = [], []
    nms_cfg_ = nms_cfg.copy()
    nms_type = nms_cfg_.pop('type', 'nms')
    nms_op = getattr(nms_wrapper, nms_type)
    bbox_cls_inds, bbox_keep_inds = [], []
    for i in range(1, num_classes):
        cls_inds = multi_scores[:, i] > score_thr
        bbox_cls_inds.append(cls_inds)
        if not cls_inds.any():
            continue
        if multi_bboxes.shape[1] == 4:
            _bboxes = multi_bboxes[cls_inds, :]
        else:
            _bboxes = multi_bboxes[cls_inds, i * 4:(i + 1) * 4]
        _scores = multi_scores[cls_inds, i]
        cls_dets = torch.cat([_bboxes, _scores[:, None]], dim=1)
        cls_dets, keep = nms_op(cls_dets, **nms_cfg_)
        bbox_keep_inds.append(keep)
        cls_labels = multi_bboxes.new_full((cls_dets.shape[0], ), i - 1, dtype=torch.long)
        bboxes.append(cls_dets)
        labels.append(cls_labels)
    if bboxes:
        bboxes = torch.cat(bboxes)
        labels = torch.cat(labels)
        if bboxes.shape[0] > max_num:
            _, inds = bboxes[:, -1].sort(descending=True)
            inds = inds[:max_num]
            bboxes = bboxes[inds]
            labels = labels[inds]
    else:
        bboxes = multi_bboxes.new_zeros((0, 5))
        labels = multi_bboxes.new_zeros((0, ), dtype=torch.long)
    return bboxes, labels, bbox_cls_inds, bbox_keep_inds


def thetaobb_nms_by_bbox_nms(multi_bboxes, multi_scores, bbox_cls_inds, bbox_keep_inds,
                             max_num=-1, out_dim_reg=5):
    """The reference's *stand-in* for rotated NMS: re-apply the horizontal keep
    indices to the rotated boxes (rbbox_nms.py:64-119).  Kept so that the
    reference behaviour of config 4 can be reproduced side by side with
    ``multiclass_rnms``.  Consumes ``bbox_keep_inds`` like the reference (:99)."""
    num_classes = multi_scores.shape[1]
    bboxes, labels = [], []
    for i in range(1, num_classes):
        cls_inds = bbox_cls_inds[i - 1]
        if not cls_inds.any():
            continue
        if multi_bboxes.shape[1] == out_dim_reg:
            _bboxes = multi_bboxes[cls_inds, :]
        else:
            _bboxes = multi_bboxes[cls_inds, i * out_dim_reg:(i + 1) * out_dim_reg]
        _scores = multi_scores[cls_inds, i]
        cls_dets = torch.cat([_bboxes, _scores[:, None]], dim=1)
        keep = bbox_keep_inds.pop(0)
        cls_dets = cls_dets[keep, :]
        cls_labels = multi_bboxes.new_full((cls_dets.shape[0], ), i - 1, dtype=torch.long)
        bboxes.append(cls_dets)
        labels.append(cls_labels)
    if bboxes:
        bboxes = torch.cat(bboxes)
        labels = torch.cat(labels)
        if bboxes.shape[0] > max_num:
            _, inds = bboxes[:, -1].sort(descending=True)
            inds = inds[:max_num]
            bboxes = bboxes[inds]
            labels = labels[inds]
    else:
        bboxes = multi_bboxes.new_zeros((0, out_dim_reg + 1))
        labels = multi_bboxes.new_zeros((0, ), dtype=torch.long)
    return bboxes, labels


def get_det_rbboxes(rois, cls_score, rbbox_pred, img_shape, scale_factor, rescale=False, cfg=None,
                    target_means=(0., 0., 0., 0., 0.), target_stds=(0.1, 0.1, 0.2, 0.2, 0.1),
                    bbox_cls_inds=None, bbox_keep_inds=None, rotated_nms=True):
    """theta-OBB detections of one image from the rbbox head's outputs.

    Counterpart of ``RBBoxHead.get_det_rbboxes_parallel``
    (mmdet/models/bbox_heads/rbbox_head.py:253-296, encode='thetaobb'): softmax,
    ``delta2thetaobb`` with the head's means/stds, optional rescale, then NMS.

    ``rotated_nms=True`` runs the call the reference left commented out (:294-295):
    ``multiclass_rnms(thetaobbs, scores, cfg.score_thr, polygon_nms_iou_thr,
    max_per_img)``.  ``rotated_nms=False`` reproduces the reference's behaviour, i.e.
    re-applies the horizontal keep indices (:287-293) -- then ``bbox_cls_inds`` /
    ``bbox_keep_inds`` from ``multiclass_nms_with_index`` are required.

    ``cfg`` needs ``score_thr``, ``max_per_img`` and ``polygon_nms_iou_thr`` (attribute or
    key access), as test_cfg.rcnn of configs/dota/dota_v002_theta_obb_r50_v1_train.py:125-132.
    """
    import torch.nn.functional as F

    from .rbbox_transforms import delta2thetaobb, thetaobb_rescale

    def get(name):
        return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)

    if isinstance(cls_score, list):
        cls_score = sum(cls_score) / float(len(cls_score))
    scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
    if rbbox_pred is not None:
        rbboxes = delta2thetaobb(rois[:, 1:], rbbox_pred, target_means, target_stds, img_shape)
    else:
        rbboxes = rois[:, 1:]
    if rescale:
        rbboxes = thetaobb_rescale(rbboxes, scale_factor, reverse_flag=True)
    if cfg is None:
        return rbboxes, scores
    if rotated_nms:
        return multiclass_rnms(rbboxes, scores, get('score_thr'),
                               dict(type='rnms', iou_thr=get('polygon_nms_iou_thr')),
                               get('max_per_img'))
    return thetaobb_nms_by_bbox_nms(rbboxes, scores, bbox_cls_inds, bbox_keep_inds,
                                    get('max_per_img'), out_dim_reg=5)
