"""Rotated-box overlaps: the theta-OBB / polygon counterpart of
``bbox_overlaps`` (reference: mmdet/core/bbox/geometry.py:4-88), computed by
the HIP kernel behind ``obb_overlaps_f32`` (include/obbops.h)."""
import ctypes

import torch

from . import _lib


def _hbb_overlaps_torch(b1, b2, mode, is_aligned):
    """geometry.py:57-86 as differentiable tensor expressions: only used for the BACKWARD of
    ``bbox_overlaps`` (the forward values come from the kernel)."""
    area1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    area2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    if is_aligned:
        lt, rb = torch.max(b1[:, :2], b2[:, :2]), torch.min(b1[:, 2:], b2[:, 2:])
    else:
        lt, rb = torch.max(b1[:, None, :2], b2[:, :2]), torch.min(b1[:, None, 2:], b2[:, 2:])
        area1 = area1[:, None]
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    return overlap / (area1 + area2 - overlap) if mode == 'iou' else overlap / area1


class _HbbOverlapsGrad(torch.autograd.Function):
    """``bbox_overlaps`` with a gradient, so that it also serves the reference's IoU loss
    (models/losses/iou_loss.py:25 calls ``bbox_overlaps(pred, target, is_aligned=True)`` and
    differentiates through it).  Forward: the HIP kernel.  Backward: autograd of the reference's tensor
    expressions, re-evaluated on the saved boxes."""

    @staticmethod
    def forward(ctx, b1, b2, mode, is_aligned):
        ctx.save_for_backward(b1, b2)
        ctx.args = (mode, is_aligned)
        with torch.no_grad():
            return _overlaps(b1, b2, mode, is_aligned, 4, _lib.KIND_HBB)

    @staticmethod
    def backward(ctx, grad_out):
        b1, b2 = ctx.saved_tensors
        mode, is_aligned = ctx.args
        with torch.enable_grad():
            x1 = b1.detach().float().requires_grad_(True)
            x2 = b2.detach().float().requires_grad_(True)
            out = _hbb_overlaps_torch(x1, x2, mode, is_aligned)
            g1, g2 = torch.autograd.grad(out, [x1, x2], grad_out.reshape(out.shape).to(out.dtype), allow_unused=True)
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        return (g1.to(b1.dtype) if need1 and g1 is not None else None,
                g2.to(b2.dtype) if need2 and g2 is not None else None, None, None)


def _overlaps(b1, b2, mode, is_aligned, cols, kind):
    assert mode in ['iou', 'iof']
    if not isinstance(b1, torch.Tensor) or not isinstance(b2, torch.Tensor):
        raise TypeError('boxes must be torch.Tensor, got {} and {}'.format(type(b1), type(b2)))
    if b1.numel() == 0:
        b1 = b1.reshape(0, cols)
    if b2.numel() == 0:
        b2 = b2.reshape(0, cols)
    assert b1.dim() == 2 and b1.size(1) == cols, 'expected (m, %d) boxes' % cols
    assert b2.dim() == 2 and b2.size(1) == cols, 'expected (n, %d) boxes' % cols
    rows, n = b1.size(0), b2.size(0)
    if is_aligned:
        assert rows == n
    # empty cases keep the reference's shapes (geometry.py:54-55); values there
    # are uninitialised in the reference, zeros here.
    if rows * n == 0:
        return b1.new_zeros((rows, 1) if is_aligned else (rows, n))
    if torch.is_grad_enabled() and (b1.requires_grad or b2.requires_grad):
        if kind != _lib.KIND_HBB:
            raise RuntimeError('rbbox_overlaps / pointobb_overlaps have no gradient: detach the boxes '
                               '(target assignment) or use bbox_overlaps for a horizontal IoU loss')
        return _HbbOverlapsGrad.apply(b1, b2, mode, is_aligned)
    a = _lib.to_device_f32(b1)
    b = _lib.to_device_f32(b2).to(a.device)
    out = torch.empty((rows,) if is_aligned else (rows, n), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_overlaps_f32(
            ctypes.c_void_p(a.data_ptr()), rows, ctypes.c_void_p(b.data_ptr()), n, kind,
            _lib.MODE[mode], 1 if is_aligned else 0, ctypes.c_void_p(out.data_ptr()),
            _lib.stream_ptr(a.device))
    _lib.check(rc)
    return out.to(device=b1.device, dtype=b1.dtype if b1.is_floating_point() else torch.float32)


def rbbox_overlaps(rbboxes1, rbboxes2, mode='iou', is_aligned=False):
    """Overlap between two sets of theta-OBBs.

    Same calling shape as the reference's ``bbox_overlaps`` (geometry.py:4).

    Args:
        rbboxes1 (Tensor): (m, 5) in <cx, cy, w, h, theta> (radians; the
            cv2.boxPoints meaning of mmdet/core/rbbox/transforms.py:45-55).
        rbboxes2 (Tensor): (n, 5); if ``is_aligned`` then m == n.
        mode (str): "iou" or "iof" (intersection over the area of rbboxes1).
        is_aligned (bool): pairwise instead of all-pairs.

    Returns:
        Tensor: (m, n), or (m,) when aligned ((m, 1) for empty aligned input,
        as geometry.py:54-55).  Computed in fp32 on the HIP device; returned on
        the input's device in the input's dtype.

    Example:
        >>> a = torch.tensor([[0., 0., 1., 1., 0.]]).cuda()
        >>> b = torch.tensor([[0., 0., 1., 1., 0.785398163]]).cuda()
        >>> rbbox_overlaps(a, b)   # regular octagon: sqrt(2)/2
        tensor([[0.7071]], device='cuda:0')
    """
    return _overlaps(rbboxes1, rbboxes2, mode, is_aligned, 5, _lib.KIND_RBOX)


def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False):
    """Overlap between two sets of HORIZONTAL boxes: the reference's ``bbox_overlaps``
    (mmdet/core/bbox/geometry.py:4-88) with the same signature, return shapes and values
    -- the kernel's horizontal kind evaluates geometry.py:61-86 in its tensor-op order
    (+1 widths; the -0 / NaN results of degenerate boxes included), so results are
    bit-equal to the reference's fp32 CPU result.

    Args:
        bboxes1 (Tensor): (m, 4) <x1, y1, x2, y2>; bboxes2 (Tensor): (n, 4).
        mode (str): "iou" or "iof" (intersection over the area of bboxes1).

    Example (geometry.py:21-35):
        >>> bboxes1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]]).cuda()
        >>> bboxes2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]]).cuda()
        >>> bbox_overlaps(bboxes1, bboxes2)
        tensor([[0.5238, 0.0500, 0.0041],
                [0.0323, 0.0452, 1.0000],
                [0.0000, 0.0000, 0.0000]], device='cuda:0')
    """
    return _overlaps(bboxes1, bboxes2, mode, is_aligned, 4, _lib.KIND_HBB)


def pointobb_overlaps(polys1, polys2, mode='iou', is_aligned=False):
    """Same for convex quadrilaterals (m, 8) <x1, y1, ..., x4, y4> (pointobb)."""
    return _overlaps(polys1, polys2, mode, is_aligned, 8, _lib.KIND_POLY)


def thetaobb2pointobb(thetaobbs):
    """(n, 5) theta-OBBs -> (n, 8) corner points in cv2.boxPoints order.

    Batched, on-device counterpart of the per-box host routine
    mmdet/core/rbbox/transforms.py:45-55.
    """
    if not isinstance(thetaobbs, torch.Tensor):
        raise TypeError('thetaobbs must be a torch.Tensor')
    t = thetaobbs.reshape(-1, 5)
    n = t.size(0)
    if n == 0:
        return thetaobbs.new_zeros((0, 8))
    a = _lib.to_device_f32(t)
    out = torch.empty((n, 8), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().obb_rbox_to_poly_f32(ctypes.c_void_p(a.data_ptr()), n,
                                             ctypes.c_void_p(out.data_ptr()),
                                             _lib.stream_ptr(a.device))
    _lib.check(rc)
    return out.to(device=thetaobbs.device, dtype=thetaobbs.dtype)
