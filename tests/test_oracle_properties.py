"""Size-independent properties of the CPU oracle (the checker of the GPU parity tests): a bug in
the restatement itself would show here without any reference implementation of rotated IoU."""
import numpy as np
import pytest

import oracle


def _boxes(n, canvas, seed):
    return oracle.synth_boxes(n, canvas, seed)


def test_iou_range_self_and_symmetry():
    a = _boxes(400, 300, 1)[:, :5]
    b = _boxes(300, 300, 2)[:, :5]
    iou = oracle.overlaps(a, b)
    assert iou.min() >= 0.0 and iou.max() <= 1.0
    # a box against itself: 1 up to the rounding of the clip (and never above: the spec clamps)
    self_iou = np.diagonal(oracle.overlaps(a, a))
    assert self_iou.max() <= 1.0 and self_iou.min() >= 1.0 - 2e-6
    # symmetry up to the rounding of the asymmetric evaluation (subject = first box)
    assert np.abs(iou - oracle.overlaps(b, a).T).max() <= 2e-5
    # iof(a, b) * area(a) == iof(b, a) * area(b) == intersection
    ia = oracle.overlaps(a, b, 'iof') * (a[:, 2] * a[:, 3])[:, None]
    ib = oracle.overlaps(b, a, 'iof').T * (b[:, 2] * b[:, 3])[None, :]
    assert np.abs(ia - ib).max() <= 2e-2 * max(1.0, np.abs(ia).max()) * 1e-2


def test_iou_fp32_spec_tracks_the_fp64_algorithm():
    a = _boxes(300, 200, 3)[:, :5]
    b = _boxes(300, 200, 4)[:, :5]
    for mode in ('iou', 'iof'):
        assert np.abs(oracle.overlaps(a, b, mode) - oracle.overlaps_f64(a, b, mode)).max() <= 1e-5


def test_iou_translation_and_half_turn_invariance():
    a = _boxes(200, 200, 5)[:, :5]
    b = _boxes(200, 200, 6)[:, :5]
    base = oracle.overlaps(a, b)
    sh = np.array([512.0, -256.0, 0, 0, 0], np.float32)        # exactly representable shift
    assert np.abs(oracle.overlaps(a + sh, b + sh) - base).max() <= 1e-5
    turn = np.array([0, 0, 0, 0, np.pi], np.float32)             # the same rectangle, half a turn on
    assert np.abs(oracle.overlaps(a + turn, b) - base).max() <= 1e-5
    swap = a.copy()                                             # (w, h, theta) == (h, w, theta + pi/2)
    swap[:, [2, 3]] = a[:, [3, 2]]
    swap[:, 4] = a[:, 4] + np.float32(np.pi / 2)
    assert np.abs(oracle.overlaps(swap, b) - base).max() <= 1e-5


def test_nms_idempotent_monotone_and_grouping():
    d = _boxes(3000, 400, 7)
    k1 = oracle.rnms(d, 0.4)
    # survivors do not suppress each other
    assert np.array_equal(oracle.rnms(d[k1], 0.4), np.arange(len(k1)))
    iou = oracle.overlaps(d[k1, :5], d[k1, :5])
    np.fill_diagonal(iou, 0.0)
    assert iou.max() < 0.4
    # every suppressed box overlaps a KEPT box with a higher score (or equal score, lower index)
    kept = np.zeros(len(d), bool); kept[k1] = True
    sup = np.nonzero(~kept)[0]
    ov = oracle.overlaps(d[sup, :5], d[k1, :5])
    better = (d[k1, 5][None, :] > d[sup, 5][:, None]) | ((d[k1, 5][None, :] == d[sup, 5][:, None]) & (k1[None, :] < sup[:, None]))
    assert ((ov >= 0.4) & better).any(axis=1).all()
    # a higher threshold keeps a superset?  (not in general for greedy NMS) -- but thr > 1 keeps everything
    assert len(oracle.rnms(d, 1.01)) == len(d)
    # grouped NMS == per-group NMS
    g = np.random.default_rng(0).integers(0, 4, len(d)).astype(np.int32)
    kg = oracle.rnms(d, 0.4, groups=g)
    parts = [np.nonzero(g == c)[0][oracle.rnms(d[g == c], 0.4)] for c in range(4)]
    assert np.array_equal(kg, np.sort(np.concatenate(parts)))


def test_polygon_and_rbox_encodings_agree():
    d = _boxes(1500, 300, 8)
    p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
    # the corner encoding is rounded to fp32 once more: keep sets agree on all but razor-edge pairs
    kr, kp = oracle.rnms(d, 0.5), oracle.poly_nms(p, 0.5)
    assert len(np.setxor1d(kr, kp)) <= 2


def test_rroi_align_constant_map_and_linearity():
    rng = np.random.default_rng(9)
    rois = np.concatenate([np.zeros((30, 1), np.float32), _boxes(30, 120, 10)[:, :5]], 1)
    rois[:, 1:3] = rng.uniform(40, 80, (30, 2))               # well inside the 32 x 32 map at scale 1/4
    rois[:, 3:5] = rng.uniform(8, 40, (30, 2))
    ones = np.ones((1, 3, 32, 32), np.float32)
    out = oracle.rroi_align_forward(ones, rois, 7, 0.25, 2)
    assert np.abs(out - 1.0).max() <= 1e-6                     # bilinear weights sum to one
    f1 = rng.standard_normal((1, 3, 32, 32)).astype(np.float32)
    f2 = rng.standard_normal((1, 3, 32, 32)).astype(np.float32)
    lhs = oracle.rroi_align_forward(f1 + 2.0 * f2, rois, 7, 0.25, 2)
    rhs = oracle.rroi_align_forward(f1, rois, 7, 0.25, 2) + 2.0 * oracle.rroi_align_forward(f2, rois, 7, 0.25, 2)
    assert np.abs(lhs - rhs).max() <= 1e-4
    # backward is the adjoint of forward: <forward(f), g> == <f, backward(g)>
    g = rng.standard_normal(out.shape).astype(np.float32)
    adj = oracle.rroi_align_backward(g, rois, f1.shape, 0.25, 2)
    assert abs(float((oracle.rroi_align_forward(f1, rois, 7, 0.25, 2) * g).sum()) - float((f1 * adj).sum())) <= 1e-2
