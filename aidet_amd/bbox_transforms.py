"""Horizontal-box codec and result plumbing (host side, pure torch).

Counterparts of mmdet/core/bbox/transforms.py: ``delta2bbox`` (:34-111), ``bbox2roi``
(:149-168), ``bbox2result`` (:181-199).  They sit between the network outputs and the
NMS / RoI-Align kernels of config 4 (rpn_head.py:86-87, bbox_head.py:149-151) and are pinned
by values captured from the reference (tests/golden/detector_parts.npz).
"""
import math

import numpy as np
import torch


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None,
               wh_ratio_clip=16 / 1000):
    """Apply network deltas to anchors / proposals.

    rois (N, 4) xyxy, deltas (N, 4k) -> boxes (N, 4k).  +1 pixel widths, centre shift by
    ``pw * dx`` (one fused multiply-add, like the reference's ``addcmul``), log-size clamp at
    |log(wh_ratio_clip)|, optional clamp to the image (``max_shape`` = (H, W, ...)).

    Example (transforms.py:64-76):
        >>> rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
        >>> deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
        >>> delta2bbox(rois, deltas, max_shape=(32, 32))
        tensor([[0.0000, 0.0000, 1.0000, 1.0000],
                [0.2817, 0.2817, 4.7183, 4.7183],
                [0.0000, 0.6321, 7.3891, 0.3679],
                [5.8967, 2.9251, 5.5033, 3.2749]])
    """
    k = deltas.size(1) // 4
    d = deltas.view(deltas.size(0), k, 4) * deltas.new_tensor(stds) + deltas.new_tensor(means)
    max_ratio = abs(math.log(wh_ratio_clip))
    dx, dy = d[..., 0], d[..., 1]
    dw, dh = d[..., 2].clamp(-max_ratio, max_ratio), d[..., 3].clamp(-max_ratio, max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5)[:, None]
    py = ((rois[:, 1] + rois[:, 3]) * 0.5)[:, None]
    pw = (rois[:, 2] - rois[:, 0] + 1.0)[:, None]
    ph = (rois[:, 3] - rois[:, 1] + 1.0)[:, None]
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx = torch.addcmul(px.expand_as(dx), pw.expand_as(dx), dx)
    gy = torch.addcmul(py.expand_as(dy), ph.expand_as(dy), dy)
    x1, y1 = gx - gw * 0.5 + 0.5, gy - gh * 0.5 + 0.5
    x2, y2 = gx + gw * 0.5 - 0.5, gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1, x2 = x1.clamp(0, max_shape[1] - 1), x2.clamp(0, max_shape[1] - 1)
        y1, y2 = y1.clamp(0, max_shape[0] - 1), y2.clamp(0, max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1).view_as(deltas)


def bbox2roi(bbox_list):
    """list of per-image (n_i, 4+) boxes -> (sum n_i, 5) [batch_index, x1, y1, x2, y2]."""
    rois = [torch.cat([b.new_full((b.size(0), 1), i), b[:, :4]], dim=-1) if b.size(0) > 0
            else b.new_zeros((0, 5)) for i, b in enumerate(bbox_list)]
    return torch.cat(rois, 0)


def bbox2result(bboxes, labels, num_classes):
    """(n, 5) detections + (n,) 0-based labels -> list of num_classes - 1 numpy arrays
    (``num_classes`` counts the background, as in the reference)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, bboxes.shape[1] if bboxes.dim() == 2 else 5), dtype=np.float32)
                for _ in range(num_classes - 1)]
    b, l = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [b[l == i, :] for i in range(num_classes - 1)]


def bbox_flip(bboxes, img_shape):
    """Mirror (..., 4k) xyxy boxes left-right inside an image of width ``img_shape[1]``
    (transforms.py:114-131; the +1 pixel convention: x -> W - x - 1, with x1 / x2 swapping roles).
    Tensor or ndarray in, same type out."""
    assert bboxes.shape[-1] % 4 == 0
    w = img_shape[1]
    out = bboxes.clone() if isinstance(bboxes, torch.Tensor) else np.array(bboxes, copy=True)
    out[..., 0::4] = w - bboxes[..., 2::4] - 1
    out[..., 2::4] = w - bboxes[..., 0::4] - 1
    return out


def bbox_mapping(bboxes, img_shape, scale_factor, flip):
    """Original image scale -> a test-time augmentation's scale (transforms.py:134-139):
    scale first, then flip inside the augmented image."""
    scaled = bboxes * scale_factor
    return bbox_flip(scaled, img_shape) if flip else scaled


def bbox_mapping_back(bboxes, img_shape, scale_factor, flip):
    """The inverse (transforms.py:142-146): un-flip inside the augmented image, then divide."""
    return (bbox_flip(bboxes, img_shape) if flip else bboxes) / scale_factor
