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


def rbox_corners(b):
    """(n, 5) theta-OBBs -> (n, 4, 2) corners (tensor expressions; the cv2.boxPoints meaning of the angle)."""
    cx, cy, w, h, th = b.unbind(-1)
    c, s = torch.cos(th), torch.sin(th)
    lx = torch.stack([-w, w, w, -w], -1) * 0.5
    ly = torch.stack([-h, -h, h, h], -1) * 0.5
    return torch.stack([cx[:, None] + lx * c[:, None] - ly * s[:, None],
                        cy[:, None] + lx * s[:, None] + ly * c[:, None]], -1)


def _cross2(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def _quad_intersection_area(P, Q):
    """Area of the intersection of two convex quads (n, 4, 2) with the SAME winding, as differentiable tensor
    expressions: the vertices of the intersection are the corners of one quad inside the other (4 + 4) and
    the crossings of their sides (16); the valid ones are ordered by angle about their centroid and the
    shoelace formula sums them (invalid slots repeat the first vertex: zero area)."""
    n = P.size(0)
    org = P.mean(1, keepdim=True).detach()             # coordinates relative to the first box: no cancellation
    P, Q = P - org, Q - org
    eps = 1e-12 if P.dtype == torch.float64 else 1e-7

    def inside(pts, quad):                             # pts (n, 4, 2) in quad (n, 4, 2): same side of all 4 edges
        a = quad[:, None, :, :]                        # (n, 1, 4, 2)
        e = torch.roll(quad, -1, 1)[:, None, :, :] - a
        cr = _cross2(e, pts[:, :, None, :] - a)        # (n, 4 pts, 4 edges)
        sgn = _cross2(torch.roll(quad, -1, 1) - quad, torch.roll(quad, -2, 1) - torch.roll(quad, -1, 1)).sum(1)  # winding
        cr = cr * torch.sign(sgn)[:, None, None]
        return (cr >= -eps).all(-1)
    p_in, q_in = inside(P, Q), inside(Q, P)
    r = (torch.roll(P, -1, 1) - P)[:, :, None, :]      # (n, 4, 1, 2) sides of P
    sd = (torch.roll(Q, -1, 1) - Q)[:, None, :, :]     # (n, 1, 4, 2) sides of Q
    qp = Q[:, None, :, :] - P[:, :, None, :]           # (n, 4, 4, 2)
    den = _cross2(r, sd)
    ok = den.abs() > eps
    den_s = torch.where(ok, den, torch.ones_like(den))
    t = _cross2(qp, sd) / den_s
    u = _cross2(qp, r) / den_s
    hit = ok & (t >= 0) & (t <= 1) & (u >= 0) & (u <= 1)
    x = P[:, :, None, :] + t[..., None] * r            # (n, 4, 4, 2)
    pts = torch.cat([P, Q, x.reshape(n, 16, 2)], 1)    # (n, 24, 2)
    valid = torch.cat([p_in, q_in, hit.reshape(n, 16)], 1)
    cnt = valid.sum(1, keepdim=True).clamp(min=1)
    ctr = (pts * valid[..., None]).sum(1, keepdim=True) / cnt[..., None]
    d = (pts - ctr).detach()
    ang = torch.atan2(d[..., 1], d[..., 0])
    ang = torch.where(valid, ang, torch.full_like(ang, 1e9))
    order = ang.argsort(1)
    srt = torch.gather(pts, 1, order[..., None].expand(-1, -1, 2))
    vs = torch.gather(valid, 1, order)
    srt = torch.where(vs[..., None], srt, srt[:, :1])  # invalid slots: the first vertex (adds no area)
    area = 0.5 * _cross2(srt, torch.roll(srt, -1, 1)).sum(1).abs()
    return torch.where(valid.sum(1) >= 3, area, torch.zeros_like(area))


def rbbox_overlaps_torch(b1, b2, mode='iou', is_aligned=False):
    """Rotated IoU / IoF as DIFFERENTIABLE tensor expressions (any device, fp32 or fp64): the second algorithm
    behind the gradient of ``rbbox_overlaps`` (the reference's IoU loss differentiates through the aligned
    horizontal IoU, models/losses/iou_loss.py:25; this is the rotated counterpart).  Values agree with the
    kernel's to rounding; the kernel, not this, produces the forward values of ``rbbox_overlaps``."""
    assert mode in ('iou', 'iof')
    if not is_aligned:
        m, n = b1.size(0), b2.size(0)
        b1 = b1[:, None, :].expand(m, n, 5).reshape(-1, 5)
        b2 = b2[None, :, :].expand(m, n, 5).reshape(-1, 5)
    inter = _quad_intersection_area(rbox_corners(b1), rbox_corners(b2))
    a1, a2 = b1[:, 2] * b1[:, 3], b2[:, 2] * b2[:, 3]
    inter = torch.min(inter, torch.min(a1, a2).clamp(min=0))
    den = a1 + a2 - inter if mode == 'iou' else a1
    out = torch.where(den > 0, inter / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))
    return out if is_aligned else out.view(m, n)


class _RboxOverlapsGrad(torch.autograd.Function):
    """``rbbox_overlaps`` with a gradient (a rotated IoU loss): forward = the HIP kernel, backward = autograd of
    ``rbbox_overlaps_torch`` re-evaluated on the saved boxes (the pattern of ``_HbbOverlapsGrad``)."""
    MAX_PAIRS = 1 << 20

    @staticmethod
    def forward(ctx, b1, b2, mode, is_aligned):
        ctx.save_for_backward(b1, b2)
        ctx.args = (mode, is_aligned)
        with torch.no_grad():
            return _overlaps(b1, b2, mode, is_aligned, 5, _lib.KIND_RBOX)

    @staticmethod
    def backward(ctx, grad_out):
        b1, b2 = ctx.saved_tensors
        mode, is_aligned = ctx.args
        with torch.enable_grad():
            x1 = b1.detach().float().requires_grad_(True)
            x2 = b2.detach().float().requires_grad_(True)
            out = rbbox_overlaps_torch(x1, x2, mode, is_aligned)
            g1, g2 = torch.autograd.grad(out, [x1, x2], grad_out.reshape(out.shape).to(out.dtype), allow_unused=True)
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        return (g1.to(b1.dtype) if need1 and g1 is not None else None,
                g2.to(b2.dtype) if need2 and g2 is not None else None, None, None)


def row_block(m, rank, world):
    """[lo, hi) of the rows rank ``rank`` of ``world`` owns when ``m`` query rows are dealt in contiguous blocks
    of ceil(m / world) (the row-block partition of ``rbbox_overlaps`` needs no communication)."""
    rank, world = int(rank), int(world)
    if not 0 <= rank < world:
        raise ValueError('row_shard=(rank, world) needs 0 <= rank < world, got ({}, {})'.format(rank, world))
    per = (int(m) + world - 1) // world
    return min(rank * per, int(m)), min((rank + 1) * per, int(m))


def _overlaps(b1, b2, mode, is_aligned, cols, kind, row_shard=None):
    assert mode in ['iou', 'iof']
    if not isinstance(b1, torch.Tensor) or not isinstance(b2, torch.Tensor):
        raise TypeError('boxes must be torch.Tensor, got {} and {}'.format(type(b1), type(b2)))
    if b1.numel() == 0:
        b1 = b1.reshape(0, cols)
    if b2.numel() == 0:
        b2 = b2.reshape(0, cols)
    assert b1.dim() == 2 and b1.size(1) == cols, 'expected (m, %d) boxes' % cols
    assert b2.dim() == 2 and b2.size(1) == cols, 'expected (n, %d) boxes' % cols
    if row_shard is not None:
        # this rank's block of query rows against ALL of b2 (aligned mode: the same block of both)
        lo, hi = row_block(b1.size(0), *row_shard)
        b1 = b1[lo:hi]
        if is_aligned:
            assert b2.size(0) >= hi
            b2 = b2[lo:hi]
    rows, n = b1.size(0), b2.size(0)
    if is_aligned:
        assert rows == n
    # empty cases keep the reference's shapes (geometry.py:54-55); values there
    # are uninitialised in the reference, zeros here.
    if rows * n == 0:
        return b1.new_zeros((rows, 1) if is_aligned else (rows, n))
    if torch.is_grad_enabled() and (b1.requires_grad or b2.requires_grad):
        if kind == _lib.KIND_HBB:
            return _HbbOverlapsGrad.apply(b1, b2, mode, is_aligned)
        if kind == _lib.KIND_RBOX:
            if rows * (1 if is_aligned else n) > _RboxOverlapsGrad.MAX_PAIRS:
                raise RuntimeError('rbbox_overlaps: a gradient over {} pairs is not supported (the backward holds 24 '
                                   'candidate vertices per pair); use is_aligned=True on matched pairs, as an IoU loss '
                                   'does'.format(rows * (1 if is_aligned else n)))
            return _RboxOverlapsGrad.apply(b1, b2, mode, is_aligned)
        raise RuntimeError('pointobb_overlaps has no gradient: detach the boxes (target assignment)')
    if kind == _lib.KIND_HBB and b1.dtype == torch.float64 and b2.dtype == torch.float64:
        # the reference's expressions follow the input's dtype (geometry.py:47-88): double in, double out
        _lib.require_gpu()
        dev = b1.device if b1.is_cuda else (b2.device if b2.is_cuda else torch.device('cuda', torch.cuda.current_device()))
        a, b = b1.detach().to(dev).contiguous(), b2.detach().to(dev).contiguous()
        out = torch.empty((rows,) if is_aligned else (rows, n), dtype=torch.float64, device=dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().obb_hbb_overlaps_f64(a.data_ptr(), rows, b.data_ptr(), n, _lib.MODE[mode], 1 if is_aligned else 0,
                                                       out.data_ptr(), _lib.current_stream_handle(dev)))
        return out.to(b1.device)
    if b1.dtype == torch.float64 and kind != _lib.KIND_HBB:
        from .nms_wrapper import _warn_downcast
        _warn_downcast('rbbox_overlaps' if kind == _lib.KIND_RBOX else 'pointobb_overlaps')
    a = _lib.to_device_f32(b1)
    b = _lib.to_device_f32(b2).to(a.device)
    out = torch.empty((rows,) if is_aligned else (rows, n), dtype=torch.float32, device=a.device)
    with _lib.on_device(a.device):
        rc = _lib.lib().obb_overlaps_f32(
            ctypes.c_void_p(a.data_ptr()), rows, ctypes.c_void_p(b.data_ptr()), n, kind,
            _lib.MODE[mode], 1 if is_aligned else 0, ctypes.c_void_p(out.data_ptr()),
            _lib.stream_ptr(a.device))
    _lib.check(rc)
    return out.to(device=b1.device, dtype=b1.dtype if b1.is_floating_point() else torch.float32)


def rbbox_overlaps(rbboxes1, rbboxes2, mode='iou', is_aligned=False, row_shard=None):
    """Overlap between two sets of theta-OBBs.

    Same calling shape as the reference's ``bbox_overlaps`` (geometry.py:4).

    Args:
        rbboxes1 (Tensor): (m, 5) in <cx, cy, w, h, theta> (radians; the
            cv2.boxPoints meaning of mmdet/core/rbbox/transforms.py:45-55).
        rbboxes2 (Tensor): (n, 5); if ``is_aligned`` then m == n.
        mode (str): "iou" or "iof" (intersection over the area of rbboxes1).
        is_aligned (bool): pairwise instead of all-pairs.
        row_shard ((rank, world), optional): compute only this rank's contiguous block of query rows
            (``row_block``) against all of ``rbboxes2`` -- the multi-GPU partition of the op: independent row
            blocks, no communication; the blocks of all ranks concatenated are the full matrix.

    Returns:
        Tensor: (m, n), or (m,) when aligned ((m, 1) for empty aligned input,
        as geometry.py:54-55).  Computed in fp32 on the HIP device; returned on
        the input's device in the input's dtype.  Differentiable w.r.t. both box sets (forward: the kernel;
        backward: autograd of ``rbbox_overlaps_torch``), so that a rotated IoU loss can be built on it the way
        the reference builds its IoU loss on ``bbox_overlaps(..., is_aligned=True)`` (losses/iou_loss.py:25).

    Example:
        >>> a = torch.tensor([[0., 0., 1., 1., 0.]]).cuda()
        >>> b = torch.tensor([[0., 0., 1., 1., 0.785398163]]).cuda()
        >>> rbbox_overlaps(a, b)   # regular octagon: sqrt(2)/2
        tensor([[0.7071]], device='cuda:0')
    """
    return _overlaps(rbboxes1, rbboxes2, mode, is_aligned, 5, _lib.KIND_RBOX, row_shard)


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
    with _lib.on_device(a.device):
        rc = _lib.lib().obb_rbox_to_poly_f32(ctypes.c_void_p(a.data_ptr()), n,
                                             ctypes.c_void_p(out.data_ptr()),
                                             _lib.stream_ptr(a.device))
    _lib.check(rc)
    return out.to(device=thetaobbs.device, dtype=thetaobbs.dtype)
