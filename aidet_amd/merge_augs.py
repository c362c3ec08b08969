"""Merging the results of test-time augmentations (multi-scale, flip).

Counterparts of mmdet/core/post_processing/merge_augs.py: ``merge_aug_proposals`` (:8-43, the RPN
leg of ``aug_test_rpn``, test_mixins.py:37-59 -- it calls ``nms``, here the HIP op),
``merge_aug_bboxes`` (:46-70) and ``merge_aug_scores`` (:73-78).  ``merge_aug_masks`` belongs to the
mask branch (out of scope); the rotated ``merge_aug_thetaobbs`` that test_mixins.py:400 calls does not
exist in the reference.  Pinned by values captured from the reference (tests/golden/detector_parts.npz).
"""
import numpy as np
import torch

from . import nms_wrapper
from .bbox_transforms import bbox_mapping_back


def _cfg(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def merge_aug_proposals(aug_proposals, img_metas, rpn_test_cfg, nms_op=None):
    """Proposals of several augmentations -> one list at the original image scale.

    Args:
        aug_proposals (list[Tensor]): (n_i, 5) x1, y1, x2, y2, score at each augmentation's scale.
        img_metas (list[dict]): ``img_shape``, ``scale_factor``, ``flip`` of each augmentation.
        rpn_test_cfg: ``nms_thr``, ``max_num``.
    Returns:
        Tensor (m, 5): the survivors of ONE NMS over all mapped-back proposals, best ``max_num`` first
        (descending score; equal scores keep the order of the NMS output).
    """
    back = []
    for props, meta in zip(aug_proposals, img_metas):
        boxes = bbox_mapping_back(props[:, :4], meta['img_shape'], meta['scale_factor'], meta['flip'])
        back.append(torch.cat([boxes, props[:, 4:5]], dim=1))
    merged, _ = (nms_op or nms_wrapper.nms)(torch.cat(back, dim=0), _cfg(rpn_test_cfg, 'nms_thr'))
    order = torch.argsort(merged[:, 4], descending=True, stable=True)
    return merged[order[:min(_cfg(rpn_test_cfg, 'max_num'), merged.shape[0])], :]


def merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, rcnn_test_cfg=None):
    """Mean of the (n, 4 * classes) boxes of all augmentations after mapping them back, and of their
    (n, classes) scores (``aug_scores`` None: boxes only).  ``img_metas[i][0]`` holds the meta of
    augmentation i (merge_augs.py:59-62)."""
    back = [bbox_mapping_back(b, m[0]['img_shape'], m[0]['scale_factor'], m[0]['flip'])
            for b, m in zip(aug_bboxes, img_metas)]
    boxes = torch.stack(back).mean(dim=0)
    if aug_scores is None:
        return boxes
    return boxes, torch.stack(aug_scores).mean(dim=0)


def merge_aug_scores(aug_scores):
    """Mean over the augmentations: tensors or ndarrays (merge_augs.py:73-78)."""
    if isinstance(aug_scores[0], torch.Tensor):
        return torch.stack(aug_scores).mean(dim=0)
    return np.mean(aug_scores, axis=0)
