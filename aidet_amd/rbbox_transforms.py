"""theta-OBB codec and geometric transforms (host side, pure torch).

Restatement of the theta-OBB parts of the reference's
``mmdet/core/rbbox/transforms.py``: ``thetaobb2delta`` (:321-353),
``delta2thetaobb`` (:356-395), ``thetaobb_flip`` (:191-203), ``thetaobb_rescale``
(:280-293), ``thetaobb_mapping(_back)`` (:398-409), ``rbbox2result`` (:615-633).
They define what the (cx, cy, w, h, theta) numbers fed to the rotated kernels
mean; the anchor angle is fixed at -pi/2 (:332,383).  Outputs are checked
against values captured from the reference (tests/golden/thetaobb_codec.npz).
"""
import math

import numpy as np
import torch


def thetaobb2delta(proposals, gt, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    """proposals (n,4) xyxy, gt (n,5) theta-OBB -> deltas (n,5)."""
    assert proposals.size(0) == gt.size(0)
    proposals = proposals.float()
    gt = gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0] + 1.0
    ph = proposals[..., 3] - proposals[..., 1] + 1.0
    ptheta = torch.full_like(px, np.float32(-np.pi / 2.0))
    gx, gy = gt[..., 0], gt[..., 1]
    gw, gh = gt[..., 2] + 1.0, gt[..., 3] + 1.0
    gtheta = gt[..., 4]
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph),
                          gtheta - ptheta], dim=-1)
    means = deltas.new_tensor(means).unsqueeze(0)
    stds = deltas.new_tensor(stds).unsqueeze(0)
    return deltas.sub_(means).div_(stds)


def delta2thetaobb(rois, deltas, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1), max_shape=None,
                   wh_ratio_clip=16 / 1000):
    """rois (n,4) xyxy, deltas (n, 5*k) -> theta-OBBs (n, 5*k)."""
    means = deltas.new_tensor(means).repeat(1, deltas.size(1) // 5)
    stds = deltas.new_tensor(stds).repeat(1, deltas.size(1) // 5)
    d = deltas * stds + means
    dx, dy, dw, dh, dtheta = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1).expand_as(dx)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1).expand_as(dy)
    pw = (rois[:, 2] - rois[:, 0] + 1.0).unsqueeze(1).expand_as(dw)
    ph = (rois[:, 3] - rois[:, 1] + 1.0).unsqueeze(1).expand_as(dh)
    ptheta = torch.full_like(dtheta, np.float32(-np.pi / 2.0))
    gw = pw * dw.exp()
    gh = ph * dh.exp()
    gx = torch.addcmul(px, pw, dx)
    gy = torch.addcmul(py, ph, dy)
    gtheta = dtheta + ptheta
    return torch.stack([gx, gy, gw, gh, gtheta], dim=-1).view_as(deltas)


def thetaobb_flip(thetaobbs, img_shape):
    """horizontal flip of (..., 5k) theta-OBBs (numpy or tensor), transforms.py:191-203."""
    assert thetaobbs.shape[-1] % 5 == 0
    w = img_shape[1]
    flipped = thetaobbs.clone() if isinstance(thetaobbs, torch.Tensor) else thetaobbs.copy()
    flipped[..., 0] = w - flipped[..., 0] - 1
    flipped[..., [3, 2]] = flipped[..., [2, 3]]
    flipped[..., 4] = -math.pi / 2.0 - flipped[..., 4]
    return flipped


def thetaobb_rescale(thetaobbs, scale_factor, reverse_flag=False):
    """scale everything but the angles; like the reference (:280-293) this modifies
    and returns its argument."""
    keep = thetaobbs.clone()
    if not reverse_flag:
        thetaobbs *= scale_factor
    else:
        thetaobbs /= scale_factor
    thetaobbs[..., 4::5] = keep[..., 4::5]
    return thetaobbs


def thetaobb_mapping(thetaobbs, img_shape, scale_factor, flip):
    new = thetaobbs * scale_factor
    return thetaobb_flip(new, img_shape) if flip else new


def thetaobb_mapping_back(thetaobbs, img_shape, scale_factor, flip):
    new = thetaobb_flip(thetaobbs, img_shape) if flip else thetaobbs
    return new / scale_factor


def rbbox2result(rbboxes, labels, num_classes):
    """(n,6|9) detections + labels -> list of per-class numpy arrays (:615-633)."""
    if rbboxes.shape[0] == 0:
        return [np.zeros((0, 6), dtype=np.float32) for _ in range(num_classes - 1)]
    rbboxes = rbboxes.cpu().numpy()
    labels = labels.cpu().numpy()
    return [rbboxes[labels == i, :] for i in range(num_classes - 1)]
