"""theta-OBB codec and geometric transforms (host side, pure torch).

Restatement of the theta-OBB parts of the reference's
``mmdet/core/rbbox/transforms.py``: ``thetaobb2delta`` (:321-353),
``delta2thetaobb`` (:356-395), ``thetaobb_flip`` (:191-203), ``thetaobb_rescale``
(:280-293), ``thetaobb_mapping(_back)`` (:398-409), ``rbbox2result`` (:615-633).
They define what the (cx, cy, w, h, theta) numbers fed to the rotated kernels
mean; the anchor angle is fixed at -pi/2 (:332,383).  Outputs are checked
against values captured from the reference (tests/golden/thetaobb_codec.npz).

The reference's ``RBBoxHead`` has two more encodings of an oriented box, selected by ``encode=``:
``pointobb`` (the four corner points, 8 numbers) and ``hobb`` (two corner points + a height).  Their
codecs -- ``pointobb2delta`` / ``delta2pointobb`` (:412-504), ``hobb2delta`` / ``delta2hobb``
(:522-599) -- and the host-side helpers around them (``pointobb2bbox`` :57-70, the two point-order rules
:93-134, ``hobb2pointobb`` :137-163, ``pointobb_flip`` :205-240, ``*_rescale`` :295-318,
``*_mapping(_back)`` :507-519, 602-613) are restated below and pinned by
tests/golden/obb_codecs.npz.

The inverse codecs the reference builds on OpenCV -- ``pointobb2thetaobb`` = ``cv2.minAreaRect`` (:30-43),
``thetaobb2pointobb`` = ``cv2.boxPoints`` (:45-55), and on top of them ``thetaobb2hobb`` (:73-91) and ``hobb_flip``
(:243-278) -- are restated WITHOUT cv2 (``min_area_rect``: rotating calipers over the edge directions of the four
points; ``box_points``: ``RotatedRect::points``), in the angle / size convention of the OpenCV releases the
reference was written against (theta in [-90, 0) degrees, the `width` is the side along (cos theta, sin theta);
an axis-aligned box therefore comes back as theta = -90 with width and height swapped).  cv2 is not part of
this image, so these four are pinned by round trips and analytic cases only (tests/test_rbbox_transforms.py),
not by vectors of the reference.  ``maskobb2thetaobb`` (:166-189; RLE decoding by pycocotools, ``cv2.findContours``,
``cv2.minAreaRect``) is restated the same way: ``rle_decode``, blobs by ``scipy.ndimage``, ``min_area_rect_points``.
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


# ---- pointobb (x1, y1, ..., x4, y4) and hobb (x1, y1, x2, y2, h) encodings ------------------------
# A proposal (xmin, ymin, xmax, ymax) anchors point k of a pointobb at its own corner k, clockwise from
# the top-left one; offsets are measured in units of the proposal's +1-pixel width / height.
_CORNER_OF = (0, 1, 2, 1, 2, 3, 0, 3)       # proposal column that anchors each pointobb coordinate


def _anchor_and_size(rois, ncoord):
    pw = rois[..., 2] - rois[..., 0] + 1.0
    ph = rois[..., 3] - rois[..., 1] + 1.0
    anchor = rois[..., list(_CORNER_OF[:ncoord])]
    size = torch.stack([pw, ph] * (ncoord // 2), dim=-1)
    return anchor, size, pw, ph


def _normalise(deltas, means, stds):
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


def pointobb2delta(proposals, gt, means=(0, 0, 0, 0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1, 1, 1, 1)):
    """proposals (n,4) xyxy, gt (n,8) corner points -> deltas (n,8): (point - corner) / (pw | ph)."""
    assert proposals.size(0) == gt.size(0)
    anchor, size, _, _ = _anchor_and_size(proposals.float(), 8)
    return _normalise((gt.float() - anchor) / size, means, stds)


def delta2pointobb(rois, deltas, means=(0, 0, 0, 0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1, 1, 1, 1), max_shape=None,
                   wh_ratio_clip=16 / 1000):
    """rois (n,4) xyxy, deltas (n, 8k) -> corner points (n, 8k).  (``max_shape`` and ``wh_ratio_clip``
    are accepted and ignored, as in the reference.)"""
    k = deltas.size(1) // 8
    d = deltas * deltas.new_tensor(stds).repeat(1, k) + deltas.new_tensor(means).repeat(1, k)
    anchor, size, _, _ = _anchor_and_size(rois, 8)
    return (size.unsqueeze(1) * d.view(-1, k, 8) + anchor.unsqueeze(1)).view_as(deltas)


def hobb2delta(proposals, gt, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    """proposals (n,4) xyxy, gt (n,5) first point, second point, height -> deltas (n,5); the height is
    encoded linearly, (h + 1 - ph) / ph."""
    assert proposals.size(0) == gt.size(0)
    gt = gt.float()
    anchor, size, _, ph = _anchor_and_size(proposals.float(), 4)
    pts = (gt[..., :4] - anchor) / size
    dh = ((gt[..., 4] + 1.0) - ph) / ph
    return _normalise(torch.cat([pts, dh.unsqueeze(-1)], dim=-1), means, stds)


def delta2hobb(rois, deltas, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1), max_shape=None, wh_ratio_clip=16 / 1000):
    """rois (n,4) xyxy, deltas (n, 5k) -> hobbs (n, 5k); the height delta is clamped to
    +-|log(wh_ratio_clip)| and decoded as ph * dh + ph (no exponential, no -1: the reference's rule)."""
    k = deltas.size(1) // 5
    d = (deltas * deltas.new_tensor(stds).repeat(1, k) + deltas.new_tensor(means).repeat(1, k)).view(-1, k, 5)
    max_ratio = np.abs(np.log(wh_ratio_clip))
    anchor, size, _, ph = _anchor_and_size(rois, 4)
    pts = size.unsqueeze(1) * d[..., :4] + anchor.unsqueeze(1)
    ph = ph.unsqueeze(1)
    gh = ph * d[..., 4].clamp(min=-max_ratio, max=max_ratio) + ph
    return torch.cat([pts, gh.unsqueeze(-1)], dim=-1).view_as(deltas)


def pointobb_rescale(pointobbs, scale_factor, reverse_flag=False):
    """scale (divide if ``reverse_flag``) every coordinate IN PLACE and return the argument (:295-306)."""
    if not reverse_flag:
        pointobbs *= scale_factor
    else:
        pointobbs /= scale_factor
    return pointobbs


def hobb_rescale(hobbs, scale_factor, reverse_flag=False):
    """same for hobbs: the two points and the height all scale (:308-318)."""
    return pointobb_rescale(hobbs, scale_factor, reverse_flag)


def pointobb2bbox(pointobb):
    """[x1, y1, ..., x4, y4] -> enclosing [xmin, ymin, xmax, ymax] (:57-70)."""
    xs, ys = pointobb[0::2], pointobb[1::2]
    return [min(xs), min(ys), max(xs), max(ys)]


def pointobb_extreme_sort(pointobb):
    """Rotate the point list so that the TOP point (smallest y; of two equally high ones the left
    one) comes first -- "top, right, bottom, left" for a clockwise quad (:93-115)."""
    pts = np.asarray(pointobb, dtype=np.float64).reshape(4, 2)
    by_y = np.argsort(pts[:, 1])
    a, b = int(by_y[0]), int(by_y[1])
    top = a if (pts[a, 1] != pts[b, 1] or pts[a, 0] < pts[b, 0]) else b
    pointobb = list(pointobb)
    return pointobb[2 * top:] + pointobb[:2 * top]


def _best_rotation(quads):
    """(n, 8) float64 quads -> each rotated so that its points lie closest (sum of squared distances)
    to the corners of its own bounding box taken clockwise from the top-left one; ties: fewest shifts."""
    xs, ys = quads[:, 0::2], quads[:, 1::2]
    x0, y0, x1, y1 = xs.min(1), ys.min(1), xs.max(1), ys.max(1)
    ref = np.stack([x0, y0, x1, y0, x1, y1, x0, y1], axis=1)
    rolled = np.stack([np.roll(quads, s, axis=1) for s in (0, 2, 4, 6)], axis=1)      # (n, 4, 8)
    dist = ((rolled - ref[:, None, :]) ** 2).sum(axis=2)
    return rolled[np.arange(len(quads)), dist.argmin(axis=1)]


def pointobb_best_point_sort(pointobb):
    """The point order the flip uses (:118-134): see ``_best_rotation``.  One quad as a list."""
    return _best_rotation(np.asarray(pointobb, dtype=np.float64).reshape(1, 8))[0].tolist()


def pointobb_flip(pointobbs, img_shape):
    """Mirror (..., 8) corner-point boxes left-right (x -> W - x - 1), restore the clockwise sense
    (points 2 and 4 trade places) and put every quad into its best-point order (:205-240).  float64 out."""
    assert pointobbs.shape[-1] % 8 == 0
    q = np.array(pointobbs, dtype=np.float64, copy=True).reshape(-1, 8)
    q[:, 0::2] = img_shape[1] - q[:, 0::2] - 1
    q = q[:, [0, 1, 6, 7, 4, 5, 2, 3]]
    out = _best_rotation(q)
    return out[0] if np.ndim(pointobbs) == 1 else out.reshape(np.shape(pointobbs))


def pointobb_mapping(pointobbs, img_shape, scale_factor, flip):
    """original image scale -> an augmentation's scale (:507-512): scale, then flip."""
    scaled = pointobbs * scale_factor
    return pointobb_flip(scaled, img_shape) if flip else scaled


def pointobb_mapping_back(pointobbs, img_shape, scale_factor, flip):
    """the inverse (:514-518): un-flip, then divide."""
    return (pointobb_flip(pointobbs, img_shape) if flip else pointobbs) / scale_factor


def hobb2pointobb(hobb):
    """[x1, y1, x2, y2, h] -> the quad whose side 1-2 is given and whose points 3, 4 lie h away from it,
    truncated to integers (:137-163)."""
    x1, y1, x2, y2, h = hobb[:5]
    turn = np.pi / 2.0 - np.arctan2(y2 - y1, x2 - x1)
    dx, dy = h * np.cos(turn), h * np.sin(turn)
    return [int(v) for v in (x1, y1, x2, y2, x2 - dx, y2 + dy, x1 - dx, y1 + dy)]


def hobb_corner_points(hobbs):
    """(n, 5) hobbs -> (n, 8) corner points, the construction of ``hobb2pointobb`` (:137-163) on tensors and
    WITHOUT its truncation to integers: points 3 and 4 lie ``h`` away from the side 1-2 along its normal
    (dx, dy)/|.| -> (-dy, dx)/|.|.  Used to give hobb detections a true polygon NMS; a degenerate side
    (both points equal) yields the normal (0, 1) of ``atan2(0, 0) = 0``."""
    x1, y1, x2, y2, h = hobbs.unbind(-1)
    turn = math.pi / 2.0 - torch.atan2(y2 - y1, x2 - x1)
    dx, dy = h * torch.cos(turn), h * torch.sin(turn)
    return torch.stack([x1, y1, x2, y2, x2 - dx, y2 + dy, x1 - dx, y1 + dy], dim=-1)


# ---- the OpenCV-backed inverse codecs, without OpenCV ---------------------------------------------------
def box_points(thetaobbs):
    """(..., 5) cx, cy, w, h, theta [rad] -> (..., 8) corners in ``cv2.boxPoints`` order
    (OpenCV ``RotatedRect::points``: a = sin/2, b = cos/2; p0 = (cx - a h - b w, cy + b h - a w),
    p1 = (cx + a h - b w, cy - b h - a w), p2 = 2c - p0, p3 = 2c - p1).  Host-side float64 twin of the
    device op ``rbbox_geometry.thetaobb2pointobb``."""
    t = np.asarray(thetaobbs, dtype=np.float64)
    cx, cy, w, h, th = (t[..., i] for i in range(5))
    a, b = np.sin(th) * 0.5, np.cos(th) * 0.5
    x0, y0 = cx - a * h - b * w, cy + b * h - a * w
    x1, y1 = cx + a * h - b * w, cy - b * h - a * w
    return np.stack([x0, y0, x1, y1, 2 * cx - x0, 2 * cy - y0, 2 * cx - x1, 2 * cy - y1], axis=-1)


def min_area_rect(quads):
    """(n, 8) four points each -> (n, 5) cx, cy, w, h, theta [DEGREES] of the smallest enclosing rectangle,
    in the convention of ``cv2.minAreaRect`` as the reference uses it (:38-40): theta in [-90, 0), `w` the
    side along (cos theta, sin theta).

    The minimal rectangle has a side collinear with a hull edge, so the six directions through two of the
    four points are tried (the four sides first, then the diagonals, which only matter for point sets that
    are not a convex quad in order); the smallest area wins, ties go to the earlier direction (OpenCV's
    choice among equal areas depends on its hull start point: a square may come back rotated by 90 degrees,
    which describes the same box).  Coincident points: a zero-length direction is skipped; all four equal
    gives w = h = 0, theta = -90 (OpenCV returns angle 0 there)."""
    q = np.asarray(quads, dtype=np.float64).reshape(-1, 4, 2)
    n = q.shape[0]
    best = np.full(n, np.inf)
    out = np.zeros((n, 5))
    out[:, 0:2] = q.mean(axis=1)
    out[:, 4] = -90.0
    for i, j in ((0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (1, 3)):
        d = q[:, j] - q[:, i]
        ln = np.hypot(d[:, 0], d[:, 1])
        ok = ln > 0
        u = np.where(ok[:, None], d / np.where(ok, ln, 1.0)[:, None], np.array([1.0, 0.0]))
        v = np.stack([-u[:, 1], u[:, 0]], axis=1)
        a = (q * u[:, None, :]).sum(-1)
        b = (q * v[:, None, :]).sum(-1)
        a0, a1, b0, b1 = a.min(1), a.max(1), b.min(1), b.max(1)
        lu, lv = a1 - a0, b1 - b0
        area = np.where(ok, lu * lv, np.inf)
        take = area < best
        if not take.any():
            continue
        best = np.where(take, area, best)
        ctr = u * ((a0 + a1) * 0.5)[:, None] + v * ((b0 + b1) * 0.5)[:, None]
        phi = np.degrees(np.arctan2(u[:, 1], u[:, 0]))
        k = np.floor(phi / 90.0) + 1.0
        theta = phi - 90.0 * k
        theta = np.where(theta >= 0.0, theta - 90.0, theta)        # (rounding at a multiple of 90)
        k = np.where(phi - 90.0 * k >= 0.0, k + 1.0, k)
        odd = (k.astype(np.int64) % 2) != 0
        w, h = np.where(odd, lv, lu), np.where(odd, lu, lv)
        new = np.stack([ctr[:, 0], ctr[:, 1], w, h, theta], axis=1)
        out = np.where(take[:, None], new, out)
    return out


def min_area_rect_points(points):
    """(m, 2) points -> (cx, cy, w, h, theta [DEGREES]) of the smallest enclosing rectangle of the set, in the convention
    of ``min_area_rect``: the directions of the convex hull's edges are tried in hull order (counter-clockwise from the
    lowest-leftmost point), the smallest area wins, ties go to the earlier edge.  Fewer than two distinct points:
    w = h = 0, theta = -90; collinear points: a degenerate rectangle along them."""
    pts = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 2), axis=0)        # (sorted by x, then y)
    if len(pts) == 0:
        return 0.0, 0.0, 0.0, 0.0, -90.0
    if len(pts) == 1:
        return float(pts[0, 0]), float(pts[0, 1]), 0.0, 0.0, -90.0

    def half(seq):
        out = []
        for q in seq:
            while len(out) >= 2 and ((out[-1][0] - out[-2][0]) * (q[1] - out[-2][1]) - (out[-1][1] - out[-2][1]) * (q[0] - out[-2][0])) <= 0:
                out.pop()
            out.append(q)
        return out
    lower, upper = half(pts), half(pts[::-1])
    hull = np.asarray(lower[:-1] + upper[:-1], dtype=np.float64)
    best = None
    for i in range(len(hull)):
        d = hull[(i + 1) % len(hull)] - hull[i]
        ln = math.hypot(d[0], d[1])
        if ln == 0.0:
            continue
        u = d / ln
        v = np.array([-u[1], u[0]])
        a, b = hull @ u, hull @ v
        a0, a1, b0, b1 = a.min(), a.max(), b.min(), b.max()
        lu, lv = a1 - a0, b1 - b0
        if best is not None and not (lu * lv < best[0]):
            continue
        ctr = u * ((a0 + a1) * 0.5) + v * ((b0 + b1) * 0.5)
        phi = math.degrees(math.atan2(u[1], u[0]))
        k = math.floor(phi / 90.0) + 1.0
        theta = phi - 90.0 * k
        if theta >= 0.0:                                    # (rounding at a multiple of 90)
            theta -= 90.0
            k += 1.0
        w, h = (lv, lu) if int(k) % 2 != 0 else (lu, lv)
        best = (lu * lv, float(ctr[0]), float(ctr[1]), float(w), float(h), float(theta))
    return best[1:]


def rle_decode(rle):
    """COCO run-length encoding -> (h, w) uint8 mask, without pycocotools: ``{'size': [h, w], 'counts': ...}`` with the
    counts as a list (uncompressed) or as the compressed string / bytes of ``maskUtils.encode`` (5 bits per character
    from '0' on, bit 5 = more follows, sign-extended, every count from the third on stored as the difference to the one
    two before); runs alternate 0 / 1 from 0, column-major."""
    h, w = int(rle['size'][0]), int(rle['size'][1])
    counts = rle['counts']
    if isinstance(counts, (bytes, str)):
        data = counts.encode('ascii') if isinstance(counts, str) else counts
        runs, p = [], 0
        while p < len(data):
            x, k, more = 0, 0, True
            while more:
                c = data[p] - 48
                x |= (c & 0x1F) << (5 * k)
                more = bool(c & 0x20)
                p += 1
                k += 1
                if not more and (c & 0x10):
                    x |= -1 << (5 * k)
            if len(runs) > 2:
                x += runs[-2]
            runs.append(x)
        counts = runs
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += int(c)
        val ^= 1
    return flat.reshape((w, h)).T                             # (column-major runs)


def maskobb2thetaobb(maskobb):
    """Mask (COCO RLE dict, or a 2-D 0 / 1 array) -> [cx, cy, w, h, theta (rad)] of the smallest rectangle around its
    largest blob (:166-189: ``maskUtils.decode`` -> ``cv2.findContours(RETR_EXTERNAL)`` -> the contour of the largest
    ``cv2.contourArea`` -> ``cv2.minAreaRect``), without pycocotools and cv2 (neither is part of this image: like the other
    OpenCV-built codecs above this one is pinned by analytic cases, not by vectors of the reference):

    * blobs = 8-connected components, holes filled (an external contour ignores them);
    * a blob's contour = its pixels with a background 4-neighbour; ``contourArea`` of the polygon through those pixel
      centres by Pick's theorem, pixels - boundary pixels / 2 - 1 (exact for a contour that does not touch itself; a
      one-pixel-wide blob has area 0, as in OpenCV); the first blob in scan order wins among equal areas;
    * the rectangle = ``min_area_rect_points`` of the contour pixels (x = column, y = row).
    No foreground: [0, 0, 0, 0, 0]."""
    from scipy import ndimage
    mask = rle_decode(maskobb) if isinstance(maskobb, dict) else np.asarray(maskobb)
    mask = mask.astype(bool)
    lab, num = ndimage.label(mask, structure=np.ones((3, 3), dtype=bool))
    if num == 0:
        return [0, 0, 0, 0, 0]
    best_area, best_pts = -1.0, None
    four = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
    for sl_idx, sl in enumerate(ndimage.find_objects(lab), start=1):
        blob = ndimage.binary_fill_holes(lab[sl] == sl_idx)
        pad = np.pad(blob, 1)
        edge = pad & ~ndimage.binary_erosion(pad, structure=four)
        ys, xs = np.nonzero(edge[1:-1, 1:-1])
        area = max(float(blob.sum()) - len(ys) / 2.0 - 1.0, 0.0)
        if best_area < area:
            best_area = area
            best_pts = np.stack([xs + sl[1].start, ys + sl[0].start], axis=1)
    x, y, w, h, theta = min_area_rect_points(best_pts)
    return [x, y, w, h, theta * np.pi / 180.0]


def pointobb2thetaobb(pointobb, truncate=True):
    """[x1, y1, ..., x4, y4] -> [cx, cy, w, h, theta (rad)] (:30-43): the reference truncates the points to
    integers (``np.int0``) before ``cv2.minAreaRect``; ``truncate=False`` skips that (exact round trips)."""
    pts = np.asarray(pointobb, dtype=np.float64).reshape(8)
    if truncate:
        pts = np.trunc(pts)
    x, y, w, h, theta = min_area_rect(pts[None])[0]
    return [float(x), float(y), float(w), float(h), float(theta / 180.0 * np.pi)]


def thetaobb2pointobb_host(thetaobb):
    """[cx, cy, w, h, theta (rad)] -> the 8 corner coordinates, ``cv2.boxPoints`` order (:45-55)."""
    return box_points(np.asarray(thetaobb, dtype=np.float64)[:5]).tolist()


def thetaobb2hobb(thetaobb, pointobb_sort_fun):
    """theta-OBB -> hobb [x1, y1, x2, y2, h] (:73-91): corner points, put into the order of
    ``pointobb_sort_fun``; the first two are kept, h = distance of the last point from the first."""
    p = pointobb_sort_fun(thetaobb2pointobb_host(thetaobb))
    h = float(np.sqrt((p[6] - p[0]) ** 2 + (p[7] - p[1]) ** 2))
    return [p[0], p[1], p[2], p[3], h]


def hobb_flip(hobbs, img_shape):
    """Mirror (..., 5) hobbs left-right the way the reference does (:243-278, its live branch): every hobb ->
    integer corner points (``hobb2pointobb``) -> ``pointobb_flip`` -> ``pointobb2thetaobb`` (integer points,
    minimal rectangle) -> ``thetaobb2hobb`` in best-point order.  float64 (n, 5) out."""
    hobbs = np.asarray(hobbs, dtype=np.float64)
    if hobbs.ndim == 1:
        hobbs = hobbs[np.newaxis, ...]
    assert hobbs.shape[-1] % 5 == 0
    quads = np.array([hobb2pointobb(hb) for hb in hobbs.reshape(-1, 5)], dtype=np.float64)
    quads = pointobb_flip(quads, img_shape).reshape(-1, 8)
    out = [thetaobb2hobb(pointobb2thetaobb(qd), pointobb_best_point_sort) for qd in quads]
    return np.array(out, dtype=np.float64)


def hobb_mapping(hobbs, img_shape, scale_factor, flip):
    """original image scale -> an augmentation's scale (:602-607): scale, then flip."""
    scaled = hobbs * scale_factor
    return hobb_flip(scaled, img_shape) if flip else scaled


def hobb_mapping_back(hobbs, img_shape, scale_factor, flip):
    """the inverse (:609-613): un-flip, then divide."""
    return (hobb_flip(hobbs, img_shape) if flip else hobbs) / scale_factor
