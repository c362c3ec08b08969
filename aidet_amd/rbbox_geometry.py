"""Rotated-box overlaps: the theta-OBB / polygon counterpart of
``bbox_overlaps`` (reference: mmdet/core/bbox/geometry.py:4-88), computed by
the HIP kernel behind ``obb_overlaps_f32`` (include/obbops.h)."""
import ctypes

import torch

from . import _lib


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
