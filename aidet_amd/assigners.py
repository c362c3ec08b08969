"""Max-IoU target assignment for rotated boxes.

Restatement of the reference's ``MaxIoUAssigner``
(mmdet/core/bbox/assigners/max_iou_assigner.py:8-202; result container
assign_result.py) with the IoU supplied by the rotated kernel
(``rbbox_overlaps`` instead of ``bbox_overlaps``, max_iou_assigner.py:99).
The reference itself assigns with HORIZONTAL boxes even for its OBB models
(rbbox_cnn.py:155-173); this is the "next" row 8f-3 of SURVEY.md.
"""
import torch

from .rbbox_geometry import bbox_overlaps, rbbox_overlaps


class AssignResult(object):
    """Same fields as mmdet/core/bbox/assigners/assign_result.py."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    def add_gt_(self, gt_labels):
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


def assign_wrt_overlaps(overlaps, gt_labels=None, pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=.0,
                        gt_max_assign_all=True):
    """Assignment from a (k gts, n boxes) overlap matrix with the rules of
    max_iou_assigner.py:114-202: -1 = don't care, 0 = negative, i > 0 = 1-based gt index.

    The reference walks the gts in a Python loop (one device sync per gt); here the three
    rules are evaluated as whole-matrix expressions:
      1. negatives: best overlap below ``neg_iou_thr`` (or inside the ``(lo, hi)`` tuple);
      2. positives: best overlap >= ``pos_iou_thr`` -> that gt;
      3. every gt whose best overlap reaches ``min_pos_iou`` claims its best box(es); when
         several gts claim one box the LAST gt wins, as the reference's loop order implies --
         a column-wise max over (claim mask x gt number).
    """
    num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
    gt_inds = overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
    if num_gts == 0 or num_bboxes == 0:
        if num_gts == 0:
            gt_inds.zero_()
        labels = None if gt_labels is None else gt_inds.new_zeros((num_bboxes, ))
        return AssignResult(num_gts, gt_inds, overlaps.new_zeros((num_bboxes, )), labels=labels)

    best, best_gt = overlaps.max(dim=0)               # per box
    gt_best, gt_best_box = overlaps.max(dim=1)        # per gt
    if isinstance(neg_iou_thr, float):
        gt_inds[(best >= 0) & (best < neg_iou_thr)] = 0
    elif isinstance(neg_iou_thr, tuple):
        assert len(neg_iou_thr) == 2
        gt_inds[(best >= neg_iou_thr[0]) & (best < neg_iou_thr[1])] = 0
    gt_inds = torch.where(best >= pos_iou_thr, best_gt + 1, gt_inds)
    if gt_max_assign_all:
        claims = overlaps == gt_best[:, None]
    else:
        claims = torch.zeros_like(overlaps, dtype=torch.bool)
        claims[torch.arange(num_gts, device=overlaps.device), gt_best_box] = True
    claims &= (gt_best >= min_pos_iou)[:, None]
    number = torch.arange(1, num_gts + 1, device=overlaps.device)[:, None]
    claimed_by = (claims * number).max(dim=0)[0]      # 0 where no gt claims the box
    gt_inds = torch.where(claimed_by > 0, claimed_by, gt_inds)

    labels = None
    if gt_labels is not None:
        labels = torch.where(gt_inds > 0, gt_labels[(gt_inds - 1).clamp(min=0)], gt_inds.new_zeros(()))
    return AssignResult(num_gts, gt_inds, best, labels=labels)


class MaxIoUAssignerRotated(object):
    """MaxIoUAssigner over theta-OBBs (n, 5) with the rotated-IoU kernel.

    Same constructor arguments as the reference (max_iou_assigner.py:36-51);
    ``gpu_assign_thr`` is accepted and ignored (the overlaps always come from the
    HIP kernel; there is no CPU path to fall back to).
    """

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True,
                 ignore_iof_thr=-1, ignore_wrt_candidates=True, gpu_assign_thr=-1):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr

    box_dim = 5
    overlaps_fn = staticmethod(rbbox_overlaps)

    def assign(self, rbboxes, gt_rbboxes, gt_rbboxes_ignore=None, gt_labels=None):
        rbbox_overlaps = self.overlaps_fn
        rbboxes = rbboxes[:, :self.box_dim]
        overlaps = rbbox_overlaps(gt_rbboxes, rbboxes)
        if (self.ignore_iof_thr > 0 and gt_rbboxes_ignore is not None
                and gt_rbboxes_ignore.numel() > 0 and rbboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_overlaps = rbbox_overlaps(rbboxes, gt_rbboxes_ignore, mode='iof')
                ignore_max_overlaps, _ = ignore_overlaps.max(dim=1)
            else:
                ignore_overlaps = rbbox_overlaps(gt_rbboxes_ignore, rbboxes, mode='iof')
                ignore_max_overlaps, _ = ignore_overlaps.max(dim=0)
            overlaps[:, ignore_max_overlaps > self.ignore_iof_thr] = -1
        return assign_wrt_overlaps(overlaps, gt_labels, self.pos_iou_thr, self.neg_iou_thr,
                                   self.min_pos_iou, self.gt_max_assign_all)


class MaxIoUAssigner(MaxIoUAssignerRotated):
    """The reference's horizontal ``MaxIoUAssigner`` (max_iou_assigner.py:8-202) itself: same rules,
    overlaps from ``bbox_overlaps`` (the kernel's horizontal kind is bit-equal to the reference's
    geometry.py), so ``gt_inds`` / ``max_overlaps`` / labels equal the reference's exactly."""
    box_dim = 4
    overlaps_fn = staticmethod(bbox_overlaps)
