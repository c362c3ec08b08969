"""Training targets of the rbbox head.

Counterpart of mmdet/core/rbbox/rbbox_target.py:8-93: for the sampled proposals of every image
(positives first, then negatives) the class label, its weight, the regression target in the head's
encoding ('thetaobb' | 'pointobb' | 'hobb', ``cfg.encode``) and its weight.  The reference walks the
positives in a Python loop with a device-to-host copy and a ``tolist()`` per proposal
(rbbox_target.py:62-69); here the assigned ground truth is ONE gather and everything stays on the
proposals' device.  Pinned by values captured from the reference (tests/golden/obb_codecs.npz).
"""
import torch

from . import rbbox_transforms as T

_ENCODERS = {'thetaobb': T.thetaobb2delta, 'pointobb': T.pointobb2delta, 'hobb': T.hobb2delta}


def _encode_of(cfg):
    return cfg['encode'] if isinstance(cfg, dict) else cfg.encode


def rbbox_target_single(pos_proposals, neg_proposals, pos_assigned_gt_inds, gt_rbboxes, gt_labels,
                        rbbox_test_cfg, target_means, target_stds, out_dim_reg=5):
    """Targets of one image: rows [positives | negatives].

    Returns:
        labels (n,) int64 (0 = background), label_weights (n,) (1 everywhere), rbbox_targets
        (n, out_dim_reg) (zero rows for negatives), rbbox_weights (n, out_dim_reg) (1 on positives).
    """
    num_pos, num_neg = pos_proposals.size(0), neg_proposals.size(0)
    n = num_pos + num_neg
    labels = pos_proposals.new_zeros(n, dtype=torch.long)
    label_weights = pos_proposals.new_zeros(n)
    targets = pos_proposals.new_zeros(n, out_dim_reg)
    weights = pos_proposals.new_zeros(n, out_dim_reg)
    if num_pos > 0:
        idx = pos_assigned_gt_inds.to(device=pos_proposals.device, dtype=torch.long)
        gt = gt_rbboxes.to(pos_proposals.device)[idx].float()
        targets[:num_pos] = _ENCODERS[_encode_of(rbbox_test_cfg)](pos_proposals, gt, target_means, target_stds)
        labels[:num_pos] = gt_labels.to(pos_proposals.device)[idx].long()
        label_weights[:num_pos] = 1.0
        weights[:num_pos] = 1.0
    if num_neg > 0:
        label_weights[n - num_neg:] = 1.0
    return labels, label_weights, targets, weights


def rbbox_target(pos_proposals_list, neg_proposals_list, pos_assigned_gt_inds_list, gt_rbboxes_list,
                 gt_labels_list, rbbox_test_cfg, target_means, target_stds, out_dim_reg=5, concat=True):
    """``rbbox_target_single`` for every image; ``concat`` joins the images (rbbox_target.py:8-35)."""
    per_image = [rbbox_target_single(p, q, i, g, l, rbbox_test_cfg, target_means, target_stds, out_dim_reg)
                 for p, q, i, g, l in zip(pos_proposals_list, neg_proposals_list, pos_assigned_gt_inds_list,
                                          gt_rbboxes_list, gt_labels_list)]
    columns = tuple(list(c) for c in zip(*per_image))
    if concat:
        return tuple(torch.cat(c, 0) for c in columns)
    return columns
