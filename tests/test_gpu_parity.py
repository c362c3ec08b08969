"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).

Every test calls the product through its Python surface, which goes through the
C ABI (include/obbops.h) into the HIP kernels, and compares with the CPU oracle
(oracle/obb_oracle.c) on the same seeded inputs, or with the golden fixtures
generated from the reference itself (tests/golden/).

Bars (BASELINE.json north_star): NMS keep indices BIT-EXACT; IoU and RoI-Align
within 1e-4.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import hbb_to_rbox

pytestmark = pytest.mark.gpu

IOU_TOL = 1e-4  # BASELINE.json: "within 1e-4 on IoU/RoI-Align floats"


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device (no CPU path exists)'
    return torch.device('cuda:0')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


# --------------------------------------------------------------------------
# rbbox_overlaps
# --------------------------------------------------------------------------
@pytest.mark.parametrize('m,n', [(1, 1), (3, 5), (64, 64), (100, 257), (1000, 1000), (37, 4099)])
@pytest.mark.parametrize('mode', ['iou', 'iof'])
def test_overlaps_matches_oracle(m, n, mode):
    from aidet_amd import rbbox_overlaps
    a = oracle.synth_boxes(m, 512, 1 + m)[:, :5]
    b = oracle.synth_boxes(n, 512, 2 + n)[:, :5]
    got = rbbox_overlaps(t(a), t(b), mode=mode).cpu().numpy()
    ref = oracle.overlaps(a, b, mode=mode)
    assert got.shape == (m, n)
    assert np.abs(got - ref).max() <= IOU_TOL
    # same arithmetic spec on both sides: in practice bit-identical
    assert np.array_equal(got, ref)
    # and close to the independent fp64 truth
    assert np.abs(got - oracle.overlaps_f64(a, b, mode=mode)).max() <= IOU_TOL


def test_overlaps_cfg1_1k_x_1k():
    """BASELINE.json configs[0]: 1k x 1k synthetic (cx,cy,w,h,theta) fp32 boxes."""
    from aidet_amd import rbbox_overlaps
    a = oracle.synth_boxes(1000, 1024, 1)[:, :5]
    b = oracle.synth_boxes(1000, 1024, 2)[:, :5]
    got = rbbox_overlaps(t(a), t(b)).cpu().numpy()
    assert np.abs(got - oracle.overlaps(a, b)).max() <= IOU_TOL
    assert np.abs(got - oracle.overlaps_f64(a, b)).max() <= IOU_TOL
    assert (got > 0).sum() > 1000


@pytest.mark.parametrize('mode', ['iou', 'iof'])
def test_overlaps_aligned(mode):
    from aidet_amd import rbbox_overlaps
    a = oracle.synth_boxes(5000, 256, 5)[:, :5]
    b = oracle.synth_boxes(5000, 256, 6)[:, :5]
    got = rbbox_overlaps(t(a), t(b), mode=mode, is_aligned=True).cpu().numpy()
    ref = oracle.overlaps(a, b, mode=mode, is_aligned=True)
    assert got.shape == (5000,)
    assert np.abs(got - ref).max() <= IOU_TOL


def test_overlaps_polygons():
    from aidet_amd.rbbox_geometry import pointobb_overlaps
    a = oracle.synth_boxes(300, 256, 7)
    b = oracle.synth_boxes(500, 256, 8)
    pa = oracle.rbox_to_poly(a).astype(np.float32)
    pb = oracle.rbox_to_poly(b).astype(np.float32)
    pb[::2] = pb[::2].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)  # mixed orientation
    got = pointobb_overlaps(t(pa), t(pb)).cpu().numpy()
    ref = oracle.overlaps(pa, pb, kind='poly')
    assert np.abs(got - ref).max() <= IOU_TOL
    # polygons of the same rectangles: close to the rbox answer
    assert np.abs(got - oracle.overlaps_f64(a[:, :5], b[:, :5])).max() <= 1e-3


def test_overlaps_reference_doctest(golden):
    """Anchor A1: the numeric table of geometry.py:21-35 through theta=0 and
    theta=-pi/2 mappings."""
    from aidet_amd import rbbox_overlaps
    g = golden('hbb_iou.npz')
    for rot in (False, True):
        got = rbbox_overlaps(t(hbb_to_rbox(g['doc_b1'], rot)), t(hbb_to_rbox(g['doc_b2'], rot)))
        assert np.abs(got.cpu().numpy() - g['doc_iou']).max() <= 1e-6
        got = rbbox_overlaps(t(hbb_to_rbox(g['rand_b1'], rot)), t(hbb_to_rbox(g['rand_b2'], rot)))
        assert np.abs(got.cpu().numpy() - g['rand_iou']).max() <= IOU_TOL
        got = rbbox_overlaps(t(hbb_to_rbox(g['rand_b1'], rot)), t(hbb_to_rbox(g['rand_b2'], rot)), mode='iof')
        assert np.abs(got.cpu().numpy() - g['rand_iof']).max() <= IOU_TOL


def test_overlaps_analytic_and_degenerate():
    from aidet_amd import rbbox_overlaps
    unit = np.array([[0, 0, 1, 1, 0]], np.float32)
    cases = np.array([
        [0, 0, 1, 1, np.pi / 4],   # octagon: sqrt(2)/2
        [0, 0, 1, 1, 0],           # identical: 1
        [5, 5, 1, 1, 0.3],         # disjoint: 0
        [1, 0, 1, 1, 0],           # shared edge: 0
        [0, 0, 0, 1, 0],           # zero area: 0, never NaN
        [0, 0, 0.5, 0.5, 0.2],     # contained: 0.25
    ], np.float32)
    got = rbbox_overlaps(t(unit), t(cases)).cpu().numpy()[0]
    want = [np.sqrt(2) / 2, 1, 0, 0, 0, 0.25]
    assert np.isfinite(got).all()
    assert np.abs(got - np.array(want)).max() <= 1e-5


def test_overlaps_empty_shapes():
    """geometry.py:51-55 empty-input shapes."""
    from aidet_amd import rbbox_overlaps
    e = torch.zeros((0, 5), device=dev())
    ne = torch.tensor([[0., 0., 10., 9., 0.]], device=dev())
    assert tuple(rbbox_overlaps(e, ne).shape) == (0, 1)
    assert tuple(rbbox_overlaps(ne, e).shape) == (1, 0)
    assert tuple(rbbox_overlaps(e, e).shape) == (0, 0)
    assert tuple(rbbox_overlaps(e, e, is_aligned=True).shape) == (0, 1)


def test_overlaps_cpu_tensor_roundtrip():
    """CPU tensors are computed on the GPU and come back on the CPU."""
    from aidet_amd import rbbox_overlaps
    a = torch.from_numpy(oracle.synth_boxes(10, 64, 1)[:, :5])
    out = rbbox_overlaps(a, a)
    assert out.device.type == 'cpu' and out.shape == (10, 10)


# --------------------------------------------------------------------------
# rotated NMS: keep indices bit-exact
# --------------------------------------------------------------------------
@pytest.fixture(params=['auto', 'sorted', 'listed', 'grid', 'grid4'])
def nms_path(request, monkeypatch):
    """The library routes n <= 128 to the fused kernel, n <= 10240 to the direct
    (unsorted) scan and the rest to the spatially sorted pipeline, whose scan clips its
    candidates in line up to 65 536 boxes and lists them for a clip launch above; 'sorted' /
    'listed' force the pipeline with one or the other for every n above the fused limit."""
    from aidet_amd import _lib
    if request.param in ('grid', 'grid4'):             # the uniform grid wherever it applies (n >= 256; it may give a scene up)
        # 'grid': as the wrapper runs it since round 6 -- on a retained workspace (OBB_NMS_RETAIN): a call of the size class of its
        # predecessor is a three-launch one on the frame THAT scene left (the tests' scenes differ wildly: the repeat paths
        # run); 'grid4': the flag ignored, every call with its own frame launch
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
        _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
        _lib.set_tuning(_lib.TUNE_NMS_RETAIN, 0 if request.param == 'grid4' else 1)
    elif request.param != 'auto':
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
        _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, (1 << 40) if request.param == 'sorted' else 0)
        _lib.set_tuning(_lib.TUNE_NMS_GRID, 0)        # (the uniform-grid route has its own fixture value below)
    yield request.param
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, -1)
    _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, -1)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, -1)
    _lib.set_tuning(_lib.TUNE_NMS_RETAIN, -1)


@pytest.mark.parametrize('n,canvas', [(1, 10), (2, 10), (63, 64), (64, 64), (65, 64), (257, 128),
                                      (1000, 256), (1025, 256), (5000, 512), (5000, 2048),
                                      (20000, 2048), (20000, 8000)])
@pytest.mark.parametrize('thr', [0.3, 0.5])
def test_rnms_bit_exact(n, canvas, thr, nms_path):
    from aidet_amd import rnms
    d = oracle.synth_boxes(n, canvas, 100 + n)
    kept, inds = rnms(t(d), thr)
    ref = oracle.rnms(d, thr)
    got = inds.cpu().numpy()
    assert got.dtype == np.int64
    assert np.array_equal(got, ref)
    assert np.array_equal(kept.cpu().numpy(), d[ref])


_CFG2_REF = {}


def _cfg2_reference(canvas, seed):
    """the oracle's keep list of a configs[1] scene (14 s of one host core each): computed once per session"""
    if (canvas, seed) not in _CFG2_REF:
        d = oracle.synth_boxes(100000, canvas, seed)
        _CFG2_REF[(canvas, seed)] = (d, oracle.rnms(d, 0.5))
    return _CFG2_REF[(canvas, seed)]


def test_rnms_cfg2_100k_sparse_and_dense(nms_path):
    """BASELINE.json configs[1]: 100k synthetic OBBs, thr 0.5, keep bit-exact against the oracle -- on the route the
    library picks AND on every forced one (sorted / listed / grid; VERDICT r4 "weak" 9: above 20 000 boxes the forced
    routes had only met each other)."""
    from aidet_amd import _lib, rnms
    for canvas, seed in ((20000, 3), (4096, 4)):
        d, ref = _cfg2_reference(canvas, seed)
        _, inds = rnms(t(d), 0.5)
        got = inds.cpu().numpy()
        route = int(_lib.lib().obb_nms_last_route())
        assert np.array_equal(got, ref), 'canvas %d, route %s (%d)' % (canvas, nms_path, route)
        if nms_path in ('sorted', 'listed'):
            assert route == 2
        if nms_path in ('grid', 'grid4'):
            assert route in (4, 5)                       # (the dense scene crowds cells: the grid gives up, the call is repeated)
        # idempotence: NMS of the survivors keeps them all
        _, again = rnms(t(d[got]), 0.5)
        assert again.numel() == len(got)


def test_rnms_score_ties_lower_index_first(nms_path):
    from aidet_amd import rnms
    d = oracle.synth_boxes(3000, 256, 11)
    d[:, 5] = np.round(d[:, 5] * 8) / 8  # 9 distinct scores -> massive ties
    _, inds = rnms(t(d), 0.4)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.4))


def test_rnms_all_identical_boxes(nms_path):
    """Pathological: every pair overlaps (n^2/2 edges); forces the workspace retry."""
    from aidet_amd import rnms
    n = 3000
    d = np.tile(np.array([[50, 50, 20, 10, 0.3, 0]], np.float32), (n, 1))
    d[:, 5] = (np.random.default_rng(0).permutation(n) + 0.5) / n
    _, inds = rnms(t(d), 0.5)
    ref = oracle.rnms(d, 0.5)
    assert len(ref) == 1
    assert np.array_equal(inds.cpu().numpy(), ref)


def test_rnms_chain_dependency(nms_path):
    """A long suppress/keep chain: box k overlaps only k-1 and k+1 (many resolve rounds)."""
    from aidet_amd import rnms
    n = 2000
    d = np.zeros((n, 6), np.float32)
    d[:, 0] = np.arange(n) * 4.0   # 10-wide boxes every 4 px: IoU(k,k+1) = 6/14 = 0.43
    d[:, 1] = 5
    d[:, 2] = 10
    d[:, 3] = 10
    d[:, 5] = 1.0 - np.arange(n) / n
    _, inds = rnms(t(d), 0.4)
    ref = oracle.rnms(d, 0.4)
    assert np.array_equal(inds.cpu().numpy(), ref)
    assert len(ref) == n // 2


def test_rnms_threshold_is_greater_equal():
    """nms_cpu.cpp:56 suppresses on `>=`: identical boxes at thr = 1.0 -> one kept."""
    from aidet_amd import rnms, nms
    d = np.array([[5, 5, 10, 10, 0, 0.9], [5, 5, 10, 10, 0, 0.8]], np.float32)
    _, inds = rnms(t(d), 1.0)
    assert inds.cpu().tolist() == [0]


def test_poly_nms_bit_exact(nms_path):
    from aidet_amd import poly_nms
    d = oracle.synth_boxes(6000, 512, 21)
    p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
    p[::3, :8] = p[::3, :8].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)
    _, inds = poly_nms(t(p), 0.5)
    assert np.array_equal(inds.cpu().numpy(), oracle.poly_nms(p, 0.5))


def test_nms_wrapper_containers_and_dtypes():
    """Mirror of the reference's tests/test_nms.py:11-70 for the rotated op."""
    from aidet_amd import rnms
    base = np.array([[50.05, 34.15, 2.9, 4.5, 0., 0.9], [50.15, 34.1, 2.7, 3.4, 0., 0.9],
                     [37.6, 13.0, 5.6, 4.0, 0., 0.4], [37.45, 13.7, 5.5, 5.0, 0., 0.3]])
    for dets in (base.astype(np.float32), base.astype(np.float64)):
        sup, inds = rnms(dets, 0.7)
        assert isinstance(sup, np.ndarray) and dets.dtype == sup.dtype
        assert len(inds) == len(sup) == 3
        sup, inds = rnms(dets, 0.7, device_id=0)
        assert len(inds) == len(sup) == 3
    for tt in (torch.FloatTensor(base), torch.DoubleTensor(base), torch.FloatTensor(base).to(dev())):
        sup, inds = rnms(tt, 0.7)
        assert tt.dtype == sup.dtype and sup.device == tt.device and inds.device == tt.device
        assert inds.dtype == torch.int64 and len(inds) == len(sup) == 3
    sup, inds = rnms(np.zeros((0, 6), np.float32), 0.5)
    assert inds.shape == (0,) and inds.dtype == np.int64
    with pytest.raises(TypeError):
        rnms([[0, 0, 1, 1, 0, 1]], 0.5)


def test_hbb_nms_matches_reference_golden(golden):
    """Anchor A2: `nms` (theta=0 mapping onto the rotated kernel) against keep
    indices produced by the reference's own nms_cpu.cpp."""
    from aidet_amd import nms
    g = golden('hbb_nms.npz')
    for tag, thr in (('test4', 0.7), ('doc7', 0.7), ('ge', 1.0)):
        _, inds = nms(g[tag + '_dets'], thr)
        assert np.array_equal(inds, g[tag + '_keep'])
    for tag in 'abcd':
        dets, thr = g['rand_%s_dets' % tag], float(g['rand_%s_thr' % tag])
        _, inds = nms(t(dets), thr)
        assert np.array_equal(inds.cpu().numpy(), g['rand_%s_keep' % tag])


# --------------------------------------------------------------------------
# rotated RoI-Align
# --------------------------------------------------------------------------
def _rois(k, canvas, seed, batch=1):
    r = oracle.synth_boxes(k, canvas, seed)[:, :5]
    b = (np.arange(k) % batch).astype(np.float32)[:, None]
    return np.concatenate([b, r], 1).astype(np.float32)


@pytest.mark.parametrize('C,H,W,K,out,sn', [(16, 64, 64, 40, 7, 2), (3, 33, 47, 25, (5, 9), 0),
                                            (40, 32, 32, 10, 14, 1), (256, 64, 64, 64, 7, 2)])
def test_rroi_align_forward(C, H, W, K, out, sn):
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(C + K)
    feat = rng.standard_normal((2, C, H, W)).astype(np.float32)
    rois = _rois(K, 4 * W, 5, batch=2)
    layer = RRoIAlign(out, 0.25, sample_num=sn)
    got = layer(t(feat), t(rois)).cpu().numpy()
    ref = oracle.rroi_align_forward(feat, rois, out, 0.25, sn)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= IOU_TOL


def test_rroi_align_cfg3():
    """BASELINE.json configs[2]: 512 rotated RoIs, 256 x H x W fp32 map, 7x7."""
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = _rois(512, 1024, 5)
    got = RRoIAlign(7, 0.25, sample_num=2)(t(feat), t(rois)).cpu().numpy()
    ref = oracle.rroi_align_forward(feat, rois, 7, 0.25, 2)
    assert np.abs(got - ref).max() <= IOU_TOL


def test_rroi_align_theta0_equals_roi_align_v2():
    """Anchor A4: theta=0 reproduces RoIAlignForwardV2(aligned=False)
    (roi_align_kernel_v2.cu:62-128) on the equivalent xyxy RoI."""
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(1)
    feat = rng.standard_normal((1, 8, 50, 60)).astype(np.float32)
    x1 = rng.uniform(0, 150, 30); y1 = rng.uniform(0, 120, 30)
    w = rng.uniform(4, 90, 30); h = rng.uniform(4, 90, 30)
    xyxy = np.stack([np.zeros(30), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    rr = np.stack([np.zeros(30), x1 + w / 2, y1 + h / 2, w, h, np.zeros(30)], 1).astype(np.float32)
    got = RRoIAlign(7, 0.25, sample_num=2)(t(feat), t(rr)).cpu().numpy()
    ref = oracle.roi_align_forward(feat, xyxy, 7, 0.25, 2, version=2, aligned=False)
    assert np.abs(got - ref).max() <= IOU_TOL


def test_rroi_align_backward():
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(2)
    feat = rng.standard_normal((2, 24, 40, 40)).astype(np.float32)
    rois = _rois(50, 160, 9, batch=2)
    ft = t(feat).requires_grad_(True)
    out = RRoIAlign(7, 0.25, sample_num=2)(ft, t(rois))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(t(g))
    ref = oracle.rroi_align_backward(g, rois, feat.shape, 0.25, 2)
    got = ft.grad.cpu().numpy()
    # float atomics: summation order differs -> relative tolerance
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_rroi_align_errors():
    from aidet_amd import RRoIAlign
    layer = RRoIAlign(7, 0.25, 2)
    with pytest.raises(NotImplementedError):  # CPU features, like roi_align.py:41-42
        layer(torch.zeros(1, 2, 8, 8), torch.zeros(1, 6))
    with pytest.raises(AssertionError):       # rois must be (K, 6), cf. roi_align.py:136
        layer(torch.zeros(1, 2, 8, 8, device=dev()), torch.zeros(1, 5, device=dev()))
    out = layer(torch.zeros(1, 2, 8, 8, device=dev()), torch.zeros(0, 6, device=dev()))
    assert tuple(out.shape) == (0, 2, 7, 7)


def test_thetaobb2pointobb():
    from aidet_amd import thetaobb2pointobb
    d = oracle.synth_boxes(100, 512, 3)[:, :5]
    got = thetaobb2pointobb(t(d)).cpu().numpy()
    assert np.abs(got - oracle.rbox_to_poly(d)).max() <= 1e-3


# --------------------------------------------------------------------------
# grouped NMS, class-wise orchestration, merge
# --------------------------------------------------------------------------
def test_grouped_rnms_bit_exact(nms_path):
    from aidet_amd import rnms
    d = oracle.synth_boxes(8000, 400, 31)
    rng = np.random.default_rng(1)
    g = rng.integers(0, 15, len(d))
    thr_c = np.linspace(0.05, 0.75, 15).astype(np.float32)
    _, inds = rnms(t(d), 0.5, groups=torch.from_numpy(g))
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.5, groups=g))
    _, inds = rnms(t(d), 0.5, groups=g, thrs=thr_c[g])           # numpy side inputs
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.5, groups=g, thrs=thr_c[g]))
    # one group == plain NMS
    _, inds = rnms(t(d), 0.5, groups=np.zeros(len(d), np.int64))
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.5))


def test_multiclass_reference_orchestration_golden(golden):
    """Anchor A3: multiclass_nms_with_index + thetaobb_nms_by_bbox_nms
    (rbbox_nms.py:6-119) reproduce the outputs captured from the reference."""
    from aidet_amd import multiclass_nms_with_index, thetaobb_nms_by_bbox_nms
    g = golden('multiclass.npz')
    mb, ms, mr = t(g['multi_bboxes']), t(g['multi_scores']), t(g['multi_rbboxes'])
    cfg = dict(type='nms', iou_thr=float(g['iou_thr']))
    bboxes, labels, cls_inds, keep_inds = multiclass_nms_with_index(
        mb, ms, float(g['score_thr']), cfg, int(g['max_num']))
    assert [len(k) for k in keep_inds] == g['idx_keep_lens'].tolist()
    for i, k in enumerate(keep_inds):
        assert np.array_equal(k.cpu().numpy(), g['idx_keep_%d' % i])
    assert np.array_equal(bboxes.cpu().numpy(), g['idx_bboxes'])
    assert np.array_equal(labels.cpu().numpy(), g['idx_labels'])
    rb, rl = thetaobb_nms_by_bbox_nms(mr, ms, cls_inds, keep_inds, int(g['max_num']), 5)
    assert np.array_equal(rb.cpu().numpy(), g['rb_bboxes'])
    assert np.array_equal(rl.cpu().numpy(), g['rb_labels'])


def test_multiclass_rnms_equals_per_class_loop(golden):
    from aidet_amd import multiclass_rnms
    g = golden('multiclass.npz')
    mr, ms = g['multi_rbboxes'], g['multi_scores']
    n, ncls = ms.shape
    det, lab = multiclass_rnms(t(mr), t(ms), 0.05, dict(type='rnms', iou_thr=0.3), max_num=-1)
    det, lab = det.cpu().numpy(), lab.cpu().numpy()
    want = []
    for c in range(1, ncls):
        m = ms[:, c] > 0.05
        d = np.concatenate([mr[m, c * 5:(c + 1) * 5], ms[m, c:c + 1]], 1)
        want.append((d[oracle.rnms(d, 0.3)], c - 1))
    total = sum(len(w[0]) for w in want)
    assert len(det) == total
    for c in range(ncls - 1):
        got_c = det[lab == c]
        ref_c = want[c][0]
        assert got_c.shape == ref_c.shape
        assert np.array_equal(got_c[np.lexsort(got_c.T)], ref_c[np.lexsort(ref_c.T)])
    # top-k by score
    det2, lab2 = multiclass_rnms(t(mr), t(ms), 0.05, dict(type='rnms', iou_thr=0.3), max_num=50)
    assert len(det2) == 50 and np.all(np.diff(det2[:, -1].cpu().numpy()) <= 0)
    # the reference's disabled call site, positional (rbbox_head.py:294-295)
    from aidet_amd import multiclass_thetaobb_nms
    det3, lab3 = multiclass_thetaobb_nms(t(mr), t(ms), 0.05, 0.3, 50)
    assert torch.equal(det3, det2) and torch.equal(lab3, lab2)


def test_distributed_merge_on_gpu_single_process():
    """The merge path with the HIP rnms (world size 1; the 2-rank exchange itself is
    covered on CPU/gloo in tests/test_dist_merge.py)."""
    from aidet_amd import dota_merge
    from test_dist_merge import CLASS_THRS, NUM_CLASSES, make_patches, oracle_nms
    rows = np.concatenate(make_patches(40, 5), 0)
    kept, keep, _ = dota_merge.distributed_merge(t(rows), 0.3, NUM_CLASSES, class_thrs=CLASS_THRS)
    _, ref = dota_merge.merge_detections(torch.from_numpy(rows), 0.3, NUM_CLASSES, CLASS_THRS,
                                         nms_fn=oracle_nms)
    assert np.array_equal(keep.cpu().numpy(), ref.numpy())


def test_rroi_align_large_and_offmap_rois():
    """RoIs whose footprint exceeds the LDS tile (direct-gather path), RoIs staged in
    several channel passes, RoIs partly / wholly outside the map, bad batch index."""
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(11)
    feat = rng.standard_normal((2, 20, 180, 200)).astype(np.float32)
    rois = np.array([
        [0, 400, 360, 700, 640, 0.4],      # ~175 x 160 px footprint: direct path
        [1, 300, 300, 240, 200, -1.0],     # ~70 x 70: several channel passes
        [0, 100, 90, 40, 30, 0.2],         # small: one pass
        [1, -50, -40, 60, 60, 0.7],        # mostly off the map
        [0, 5000, 5000, 30, 30, 0.0],      # entirely off the map -> zeros
        [1, 790, 700, 90, 120, 2.5],       # bottom-right corner
        [0, 400, 360, 1.0, 1.0, 0.3],      # tiny (min_size clamp)
    ], np.float32)
    for sn in (2, 0):
        got = RRoIAlign(7, 0.25, sample_num=sn)(t(feat), t(rois)).cpu().numpy()
        ref = oracle.rroi_align_forward(feat, rois, 7, 0.25, sn)
        assert np.abs(got - ref).max() <= IOU_TOL
        assert np.all(got[4] == 0)
    bad = rois.copy()
    bad[2, 0] = 7  # batch index out of range -> zeros for that RoI (header contract)
    got = RRoIAlign(7, 0.25, sample_num=2)(t(feat), t(bad)).cpu().numpy()
    ref = oracle.rroi_align_forward(feat, rois, 7, 0.25, 2)
    assert np.all(got[2] == 0) and np.abs(got[[0, 1, 3, 4, 5, 6]] - ref[[0, 1, 3, 4, 5, 6]]).max() <= IOU_TOL


def test_hbb_roi_align_dropin_v1_v2():
    """RoIAlign (reference signature) over the rotated kernel vs the restated
    ROIAlignForwardV1 (roi_align_kernel.cu:64-117) and RoIAlignForwardV2 aligned
    (roi_align_kernel_v2.cu:62-128)."""
    from aidet_amd import RoIAlign
    rng = np.random.default_rng(4)
    feat = rng.standard_normal((2, 12, 48, 64)).astype(np.float32)
    k = 60
    x1 = rng.uniform(0, 200, k); y1 = rng.uniform(0, 150, k)
    w = rng.uniform(2, 120, k); h = rng.uniform(2, 100, k)
    rois = np.stack([(np.arange(k) % 2), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    for sn in (2, 0):
        got = RoIAlign(7, 0.25, sample_num=sn, aligned=False)(t(feat), t(rois)).cpu().numpy()
        ref = oracle.roi_align_forward(feat, rois, 7, 0.25, sn, version=1)
        assert np.abs(got - ref).max() <= IOU_TOL
        got = RoIAlign(7, 0.25, sample_num=sn, aligned=True)(t(feat), t(rois)).cpu().numpy()
        ref = oracle.roi_align_forward(feat, rois, 7, 0.25, sn, version=2, aligned=True)
        assert np.abs(got - ref).max() <= IOU_TOL


def test_rotated_extractor_forward():
    from aidet_amd import RotatedSingleRoIExtractor
    rng = np.random.default_rng(6)
    feats = [rng.standard_normal((1, 8, 256 // s, 256 // s)).astype(np.float32) for s in (1, 2, 4, 8)]
    ex = RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=7, sample_num=2), 8, [4, 8, 16, 32])
    r = oracle.synth_boxes(40, 1000, 2)[:, :5]
    r[:, 2:4] *= rng.uniform(1, 6, (40, 1)).astype(np.float32)
    rois = np.concatenate([np.zeros((40, 1), np.float32), r], 1)
    out = ex([t(f) for f in feats], t(rois)).cpu().numpy()
    lv = ex.map_roi_levels(torch.from_numpy(rois), 4).numpy()
    assert len(set(lv.tolist())) >= 3
    for i, s in enumerate((4, 8, 16, 32)):
        m = lv == i
        if m.any():
            ref = oracle.rroi_align_forward(feats[i], rois[m], 7, 1.0 / s, 2)
            assert np.abs(out[m] - ref).max() <= IOU_TOL


def test_rnms_non_finite_boxes_are_inert(nms_path):
    """NaN / inf boxes never suppress and are never suppressed (their overlap is 0 or
    NaN on the spec path); the GPU isolates them before the scan."""
    from aidet_amd import rnms
    d = oracle.synth_boxes(4000, 300, 77)
    bad = [5, 100, 101, 2500, 3999]
    d[5, 0] = np.nan
    d[100, 2] = np.inf
    d[101, 1] = -np.inf
    d[2500, 4] = np.nan
    d[3999, 3] = np.nan
    _, inds = rnms(t(d), 0.3)
    ref = oracle.rnms(d, 0.3)
    got = inds.cpu().numpy()
    assert np.array_equal(got, ref)
    assert set(bad) <= set(got.tolist())
    # all boxes non-finite
    z = np.full((10, 6), np.nan, np.float32)
    z[:, 5] = np.linspace(0, 1, 10)
    _, inds = rnms(t(z), 0.5)
    assert inds.cpu().tolist() == list(range(10))


def test_rnms_degenerate_and_huge_coordinates(nms_path):
    from aidet_amd import rnms
    rng = np.random.default_rng(8)
    d = oracle.synth_boxes(3000, 200, 9)
    d[::7, 2] = 0.0                       # zero width
    d[::11, 3] = 0.0                      # zero height
    d[::13, :2] += 1.0e6                  # far away cluster
    d[::17, 4] += 100 * np.pi             # many turns
    d[1500:1600] = d[1400:1500]           # exact duplicates (different scores)
    d[1500:1600, 5] = rng.permutation(100) / 1e4
    _, inds = rnms(t(d), 0.5)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.5))


def test_rnms_one_million_boxes_properties():
    """Size-independent checks at n = 1e6 (the O(n^2) oracle is out of reach):
    idempotence, agreement of the grouped entry point, and an exact oracle check on
    a spatial window (NMS is local: a window plus a margin decides its interior)."""
    from aidet_amd import rnms
    n, canvas = 1000000, 60000.0
    d = oracle.synth_boxes(n, canvas, 12)
    _, inds = rnms(t(d), 0.5)
    keep = inds.cpu().numpy()
    assert np.all(np.diff(keep) > 0) and 0.9 * n < len(keep) < n
    _, again = rnms(t(d[keep]), 0.5)
    assert again.numel() == len(keep)
    _, g = rnms(t(d), 0.5, groups=np.zeros(n, np.int32))
    assert np.array_equal(g.cpu().numpy(), keep)
    # exact check in a window: every box within 400 px of a window box is included, chains
    # longer than the 2000 px margin do not occur at this density
    cx, cy = d[:, 0], d[:, 1]
    sel = np.nonzero((cx > 28000) & (cx < 34000) & (cy > 28000) & (cy < 34000))[0]
    inner = (cx[sel] > 30000) & (cx[sel] < 32000) & (cy[sel] > 30000) & (cy[sel] < 32000)
    ref = sel[oracle.rnms(d[sel], 0.5)]
    kept_mask = np.zeros(n, bool); kept_mask[keep] = True
    ref_mask = np.zeros(n, bool); ref_mask[ref] = True
    assert np.array_equal(kept_mask[sel[inner]], ref_mask[sel[inner]])
    assert inner.sum() > 500


def test_get_det_rbboxes_both_modes(golden):
    """rbbox_head.py:253-296 counterpart: reference behaviour (horizontal keep re-applied)
    and the true rotated NMS the reference left commented out."""
    from aidet_amd.post_processing import get_det_rbboxes, multiclass_nms_with_index
    from aidet_amd.rbbox_transforms import delta2thetaobb
    rng = np.random.default_rng(3)
    n, ncls = 300, 6
    x1 = rng.uniform(0, 400, n); y1 = rng.uniform(0, 400, n)
    rois = np.stack([np.zeros(n), x1, y1, x1 + rng.uniform(10, 90, n), y1 + rng.uniform(10, 90, n)], 1).astype(np.float32)
    cls_score = rng.standard_normal((n, ncls)).astype(np.float32) * 2
    pred = (rng.standard_normal((n, ncls * 5)) * 0.5).astype(np.float32)
    cfg = dict(score_thr=0.05, max_per_img=100, polygon_nms_iou_thr=0.4)
    det, lab = get_det_rbboxes(t(rois), t(cls_score), t(pred), (512, 512, 3), 1.0, cfg=cfg)
    det, lab = det.cpu().numpy(), lab.cpu().numpy()
    assert det.shape[1] == 6 and len(det) == len(lab) <= 100
    # oracle: decode on CPU, per-class rnms, top-100 by score
    sc = torch.softmax(torch.from_numpy(cls_score), 1).numpy()
    rb = delta2thetaobb(torch.from_numpy(rois[:, 1:]), torch.from_numpy(pred), (0,) * 5,
                        (0.1, 0.1, 0.2, 0.2, 0.1)).numpy()
    allk = []
    for c in range(1, ncls):
        m = sc[:, c] > 0.05
        d = np.concatenate([rb[m, c * 5:(c + 1) * 5], sc[m, c:c + 1]], 1).astype(np.float32)
        allk.append(d[oracle.rnms(d, 0.4)])
    allk = np.concatenate(allk, 0)
    top = allk[np.argsort(-allk[:, 5], kind='stable')[:100]]
    assert np.allclose(np.sort(det[:, 5])[::-1], top[:, 5], atol=1e-6)
    # reference mode needs the horizontal pass first
    hb = torch.from_numpy(rois[:, 1:]).repeat(1, ncls)
    _, _, cls_inds, keep_inds = multiclass_nms_with_index(t(hb.numpy()), t(sc), 0.05, dict(type='nms', iou_thr=0.5), 100)
    det2, lab2 = get_det_rbboxes(t(rois), t(cls_score), t(pred), (512, 512, 3), 1.0, cfg=cfg,
                                 bbox_cls_inds=cls_inds, bbox_keep_inds=keep_inds, rotated_nms=False)
    assert det2.shape[1] == 6 and len(det2) <= 100


def test_rotated_max_iou_assigner_matches_reference_at_theta0(golden):
    """MaxIoUAssignerRotated on theta=0 encodings of the reference's HBBs reproduces
    the reference MaxIoUAssigner's gt_inds / labels (fixture from make_golden.py)."""
    from aidet_amd.assigners import MaxIoUAssignerRotated
    g = golden('assigner.npz')
    boxes, gts, ign = (t(hbb_to_rbox(g[k])) for k in ('boxes', 'gts', 'ign'))
    labels = t(g['labels'])
    cases = (('a', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3), False),
             ('b', dict(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.0, gt_max_assign_all=False), False),
             ('c', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5), True))
    for tag, kw, use_ign in cases:
        r = MaxIoUAssignerRotated(**kw).assign(boxes, gts, ign if use_ign else None, labels)
        assert np.array_equal(r.gt_inds.cpu().numpy(), g['assign_%s_gt_inds' % tag]), tag
        assert np.array_equal(r.labels.cpu().numpy(), g['assign_%s_labels' % tag]), tag
        assert np.abs(r.max_overlaps.cpu().numpy() - g['assign_%s_max_overlaps' % tag]).max() <= 1e-5


@pytest.mark.parametrize('n,canvas', [(1, 10), (2, 10), (64, 64), (257, 100), (700, 150), (1024, 200),
                                      (1024, 60)])
def test_rnms_fused_and_general_paths_agree(n, canvas, nms_route):
    """n <= 1024 can take the fused single-launch kernel; the tuning hooks of the library
    (obb_set_tuning) force the direct scan or the sorted pipeline instead.  All three must
    reproduce the oracle (dense canvas 60: ~1e5 edges)."""
    from aidet_amd import poly_nms, rnms
    d = oracle.synth_boxes(n, canvas, 300 + n)
    g = np.random.default_rng(n).integers(0, 3, n)
    ref = oracle.rnms(d, 0.45)
    refg = oracle.rnms(d, 0.45, groups=g)
    p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
    refp = oracle.poly_nms(p, 0.45)
    for path in ('fused', 'direct', 'sorted', 'listed'):
        nms_route(path)
        assert np.array_equal(rnms(t(d), 0.45)[1].cpu().numpy(), ref), path
        assert np.array_equal(rnms(t(d), 0.45, groups=g)[1].cpu().numpy(), refg), path
        assert np.array_equal(poly_nms(t(p), 0.45)[1].cpu().numpy(), refp), path


def test_small_nms_latency_smoke():
    """not a benchmark: the fused path must not be slower than the general one"""
    import time
    from aidet_amd import rnms
    d = t(oracle.synth_boxes(1000, 300, 5))
    for _ in range(5):
        rnms(d, 0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        rnms(d, 0.5)
    torch.cuda.synchronize()
    fused = (time.perf_counter() - t0) / 50
    assert fused < 5e-3
    print('rnms n=1000: %.1f us per call' % (fused * 1e6))


def test_general_convex_quads_gpu(nms_path):
    """pointobb_overlaps / poly_nms on convex quads that are not rectangles"""
    from aidet_amd import poly_nms
    from aidet_amd.rbbox_geometry import pointobb_overlaps
    a = oracle.synth_quads(700, 300, 31)
    b = oracle.synth_quads(900, 300, 32)
    for mode in ('iou', 'iof'):
        got = pointobb_overlaps(t(a[:, :8]), t(b[:, :8]), mode=mode).cpu().numpy()
        assert np.array_equal(got, oracle.overlaps(a[:, :8], b[:, :8], mode, kind='poly'))
        assert np.abs(got - oracle.overlaps_f64(a[:, :8], b[:, :8], mode, kind='poly')).max() <= IOU_TOL
    for n, canvas, thr in ((300, 100, 0.3), (5000, 600, 0.5), (30000, 3000, 0.4)):
        d = oracle.synth_quads(n, canvas, 40 + n)
        _, inds = poly_nms(t(d), thr)
        assert np.array_equal(inds.cpu().numpy(), oracle.poly_nms(d, thr))


@pytest.mark.parametrize('seed', range(6))
def test_rnms_randomised_shapes_and_clusters(seed, nms_path):
    """clustered scenes, extreme aspect ratios, mixed scales, random thresholds"""
    from aidet_amd import rnms
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(500, 12000))
    k = int(rng.integers(3, 40))
    centres = rng.uniform(0, 4000, (k, 2))
    which = rng.integers(0, k, n)
    d = np.zeros((n, 6), np.float32)
    d[:, :2] = centres[which] + rng.normal(0, rng.uniform(5, 200), (n, 2))
    d[:, 2] = np.exp(rng.uniform(np.log(2), np.log(400), n))
    d[:, 3] = np.exp(rng.uniform(np.log(2), np.log(60), n))
    d[:, 4] = rng.uniform(-np.pi, np.pi, n)
    d[:, 5] = rng.random(n).astype(np.float32)
    if seed % 2:
        d[:, 5] = np.round(d[:, 5], 2)          # ties
    thr = float(rng.uniform(0.1, 0.8))
    _, inds = rnms(t(d), thr)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, thr))


def test_torch_extension_matches_ctypes_path():
    """the pybind11/ATen binding and the ctypes wrappers drive the same C ABI"""
    from aidet_amd import poly_nms, rbbox_overlaps, rnms, rroi_align, torch_ext
    e = torch_ext.ext()
    d = oracle.synth_boxes(6000, 700, 91)
    g = np.random.default_rng(2).integers(0, 4, len(d)).astype(np.int32)
    dt = t(d)
    assert np.array_equal(e.rnms(dt, 0.5).cpu().numpy(), oracle.rnms(d, 0.5))
    assert torch.equal(e.rnms(dt, 0.5), rnms(dt, 0.5)[1])
    assert np.array_equal(e.rnms_grouped(dt, 0.4, t(g), None).cpu().numpy(), oracle.rnms(d, 0.4, groups=g))
    p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
    assert torch.equal(e.poly_nms(t(p), 0.5), poly_nms(t(p), 0.5)[1])
    a, b = t(d[:300, :5]), t(d[300:1000, :5])
    hb = np.concatenate([hbb := np.sort(np.random.default_rng(9).uniform(0, 300, (800, 4)), axis=1)[:, [0, 2, 1, 3]], np.random.default_rng(10).random((800, 1))], 1).astype(np.float32)
    assert np.array_equal(e.nms(t(hb), 0.5).cpu().numpy(), oracle.hbb_nms(hb, 0.5))
    assert torch.equal(e.rbbox_overlaps(a, b, 'iou', False), rbbox_overlaps(a, b))
    assert torch.equal(e.rbbox_overlaps(a, a, 'iof', True), rbbox_overlaps(a, a, mode='iof', is_aligned=True))
    assert np.abs(e.thetaobb2pointobb(a).cpu().numpy() - oracle.rbox_to_poly(d[:300])).max() <= 1e-3
    rng = np.random.default_rng(4)
    feat = t(rng.standard_normal((2, 8, 40, 40)).astype(np.float32))
    rois = np.concatenate([rng.integers(0, 2, (20, 1)).astype(np.float32), oracle.synth_boxes(20, 160, 3)[:, :5]], 1)
    out = e.rroi_align_forward(feat, t(rois), 7, 7, 0.25, 2, 1.0)
    assert torch.equal(out, rroi_align(feat, t(rois), 7, 0.25, 2))
    gin = e.rroi_align_backward(torch.ones_like(out), t(rois), 2, 8, 40, 40, 0.25, 2, 1.0)
    ref = oracle.rroi_align_backward(np.ones(out.shape, np.float32), rois, (2, 8, 40, 40), 0.25, 2)
    assert np.abs(gin.cpu().numpy() - ref).max() <= 1e-3


@pytest.mark.parametrize('case', ['one_far', 'far_cluster', 'edge_straddlers'])
def test_rnms_outliers_clamp_into_the_robust_frame(case, nms_path):
    """Boxes far outside the scene are left out of the quantisation frame of the sorted
    route and CLAMP into its edge cells (nms_quant_kernel); the keep set must not notice."""
    from aidet_amd import rnms
    n = 15000
    d = oracle.synth_boxes(n, 3000, 123)          # dense enough for a few thousand edges
    if case == 'one_far':
        d[7777, 0] = 1.0e8
        d[7777, 1] = -3.0e7
    elif case == 'far_cluster':                   # 1 % far away, overlapping each other there
        idx = np.arange(150) * 100
        d[idx, 0] = 1.0e6 + (np.arange(150) % 12) * 20.0
        d[idx, 1] = -5.0e5 + (np.arange(150) // 12) * 20.0
    else:                                         # huge boxes that cover the frame edge and beyond
        idx = np.arange(40) * 300
        d[idx, 2] = 5000.0
        d[idx, 3] = 9000.0
        d[idx[:10], 0] = -2000.0
        d[idx[10:20], 1] = 6000.0
    for thr in (0.3, 0.6):
        _, inds = rnms(t(d), thr)
        assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, thr)), (case, thr)


@pytest.mark.parametrize('case', ['staged', 'big_rois', 'adaptive_grid', 'odd_shapes'])
def test_rroi_align_backward_paths(case):
    """backward: the staged path (per-wave LDS accumulation), the direct-atomics fallback (big
    footprints, adaptive sampling grids) and RoIs that leave the map / name a bad batch index"""
    from aidet_amd import rroi_align
    rng = np.random.default_rng(11)
    if case == 'staged':
        B, C, H, W, K, out, sn, scale, canvas = 2, 40, 64, 64, 60, 7, 2, 0.25, 256
    elif case == 'big_rois':
        B, C, H, W, K, out, sn, scale, canvas = 1, 16, 96, 96, 12, 7, 2, 1.0, 96
    elif case == 'adaptive_grid':
        B, C, H, W, K, out, sn, scale, canvas = 1, 8, 48, 48, 20, 3, 0, 0.5, 96
    else:
        B, C, H, W, K, out, sn, scale, canvas = 3, 13, 37, 45, 33, (5, 9), 1, 0.25, 180
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    rois = _rois(K, canvas, 70 + K, batch=B)
    if case == 'big_rois':
        rois[:, 3:5] *= 0.6          # up to ~77 px wide at scale 1: footprints beyond the LDS quarter
    rois[0, 1:3] = [-30.0, 10.0]      # mostly off the map
    rois[1, 1:3] = [canvas + 50.0, canvas + 50.0]   # entirely off the map
    rois[2, 0] = B                     # invalid batch index: contributes nothing
    ph, pw = (out, out) if isinstance(out, int) else out
    g = rng.standard_normal((K, C, ph, pw)).astype(np.float32)
    ft = t(feat).requires_grad_(True)
    res = rroi_align(ft, t(rois), out, scale, sn)
    res.backward(t(g))
    ref = oracle.rroi_align_backward(g, rois, feat.shape, scale, sn)
    got = ft.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    # and the forward of the same call
    assert np.abs(res.detach().cpu().numpy() - oracle.rroi_align_forward(feat, rois, out, scale, sn)).max() <= IOU_TOL


@pytest.mark.parametrize('ngroups', [3, 40, 2500])
def test_grouped_rnms_many_groups_share_one_coordinate_range(ngroups, nms_path):
    """The merged NMS of tiled inference (config 5): thousands of (image, class) groups in ONE
    coordinate range.  The sorted route orders by a hash bucket of the group first (arbitrary,
    also negative, group ids); the keep set must equal the oracle's."""
    from aidet_amd import rnms
    n = 24000
    d = oracle.synth_boxes(n, 1500, 900 + ngroups)         # dense: many cross-group overlaps
    rng = np.random.default_rng(ngroups)
    ids = rng.integers(-2**31, 2**31 - 1, ngroups, dtype=np.int64).astype(np.int32)
    g = ids[rng.integers(0, ngroups, n)]
    ref = oracle.rnms(d, 0.3, groups=g)
    _, inds = rnms(t(d), 0.3, groups=t(g))
    assert np.array_equal(inds.cpu().numpy(), ref)
    # per-box thresholds on top (pruning off)
    thr = rng.choice([0.1, 0.3, 0.5], n).astype(np.float32)
    ref2 = oracle.rnms(d, 0.3, groups=g, thrs=thr)
    _, inds2 = rnms(t(d), 0.3, groups=t(g), thrs=t(thr))
    assert np.array_equal(inds2.cpu().numpy(), ref2)


def test_overlaps_16k_x_16k_properties():
    """The scale-up of configs[0]: the full 16384 x 16384 matrix on the GPU; the
    oracle checks a random sample of rows bit for bit, the rest through properties."""
    from aidet_amd import rbbox_overlaps
    m = 16384
    a = oracle.synth_boxes(m, 2048, 1)[:, :5]
    b = oracle.synth_boxes(m, 2048, 2)[:, :5]
    ta, tb = t(a), t(b)
    iou = rbbox_overlaps(ta, tb)
    assert iou.shape == (m, m)
    assert float(iou.min()) >= 0.0 and float(iou.max()) <= 1.0
    rows = np.random.default_rng(0).choice(m, 96, replace=False)
    assert np.array_equal(iou[torch.from_numpy(rows).cuda()].cpu().numpy(), oracle.overlaps(a[rows], b))
    # symmetry against the transposed problem (the subject of a clip is the first box)
    assert float((iou - rbbox_overlaps(tb, ta).t()).abs().max()) <= 2e-5
    # disjoint pairs are exact zeros: the non-zeros are exactly the pairs the oracle's prefilter lets through
    nz = int((iou > 0).sum())
    assert 0 < nz < m * m // 20
    # aligned mode = the diagonal
    assert torch.equal(rbbox_overlaps(ta, tb, is_aligned=True), iou.diagonal())


@pytest.mark.parametrize('aligned', [False, True])
def test_hbb_roi_align_uses_the_reference_expressions(aligned):
    """RoIAlign evaluates [b, x1, y1, x2, y2] with the reference's OWN expressions (V1:
    roi_align_kernel.cu:79-113, V2 aligned: roi_align_kernel_v2.cu:79-124), not through a
    mapped rotated RoI: bit for bit the oracle's V1 / V2 restatement, degenerate RoIs included,
    and within 1e-4 of the rotated op on the explicitly mapped theta = 0 RoIs."""
    from aidet_amd import rroi_align
    from aidet_amd.roi_layers import RoIAlign
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 24, 48, 56)).astype(np.float32)
    K = 70
    xy = rng.uniform(-10, 190, (K, 2)); wh = rng.uniform(2, 90, (K, 2))
    rois = np.concatenate([rng.integers(0, 2, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    rois[5, 3] = rois[5, 1] - 7.0        # x2 < x1 - 1: V1 clamps the width to 0 but keeps the start
    rois[6, 4] = rois[6, 2] - 1.0        # zero height in V1
    scale = 0.25
    b, x1, y1, x2, y2 = [t(rois[:, i].copy()) for i in range(5)]
    if not aligned:
        mapped = torch.stack([b, (x1 + x2 + 1) * 0.5, (y1 + y2 + 1) * 0.5, x2 - x1 + 1, y2 - y1 + 1, torch.zeros_like(b)], 1)
    else:
        off = 0.5 / scale
        mapped = torch.stack([b, (x1 + x2) * 0.5 - off, (y1 + y2) * 0.5 - off, x2 - x1, y2 - y1, torch.zeros_like(b)], 1)
    f1 = t(feat).requires_grad_(True)
    f2 = t(feat).requires_grad_(True)
    out1 = RoIAlign(7, scale, sample_num=2, aligned=aligned)(f1, t(rois))
    ref = oracle.roi_align_forward(feat, rois, 7, scale, 2, version=2 if aligned else 1, aligned=aligned)
    assert np.array_equal(out1.detach().cpu().numpy(), ref)
    ok = np.ones(K, bool); ok[[5, 6]] = False          # (the mapped rotated RoI is a different box there)
    out2 = rroi_align(f2, mapped, 7, scale, 2, 0.0)
    assert float((out1 - out2)[t(ok)].abs().max()) <= 1e-4
    g = t(rng.standard_normal(tuple(out1.shape)).astype(np.float32))
    g[5:7] = 0
    out1.backward(g)
    out2.backward(g)
    assert float((f1.grad - f2.grad).abs().max()) <= 1e-4 * max(1.0, float(f2.grad.abs().max()))
    # adaptive sampling grid on an empty RoI: V1 divides 0 by gh * gw = 0 -> NaN, V2 -> 0
    z = np.array([[0, 10, 10, 9, 9]], np.float32)      # (x2 + 1 - x1) = 0
    o = RoIAlign(2, 1.0, sample_num=0, aligned=aligned)(t(feat), t(z)).cpu().numpy()
    r = oracle.roi_align_forward(feat, z, 2, 1.0, 0, version=2 if aligned else 1, aligned=aligned)
    assert np.array_equal(np.isnan(o), np.isnan(r)) and np.array_equal(np.nan_to_num(o), np.nan_to_num(r))


# --------------------------------------------------------------------------
# round 2: dedicated HBB narrow phase, strict compare, Soft-NMS, async calls
# --------------------------------------------------------------------------
@pytest.mark.parametrize('route', ['auto', 'fused', 'direct', 'sorted', 'listed'])
def test_hbb_nms_exact_ties_match_reference(golden, route, nms_route):
    """Integer boxes: IoU == thr happens and nms_cpu.cpp:56 suppresses on `>=`.  The HBB rows
    are decided by the reference's own arithmetic in the kernel (OBB_KIND_HBB narrow phase)."""
    from aidet_amd import nms
    g = golden('hbb_nms_ties.npz')
    nms_route(route)
    assert nms(g['half_dets'], 0.5)[1].tolist() == g['half_keep'].tolist() == [0, 2]
    assert nms(g['half_dets'], 0.5, strict=True)[1].tolist() == [0, 1]       # nms_kernel.cu:61 `>`
    for tag in 'abc':
        d, thr, keep = g['int_%s_dets' % tag], float(g['int_%s_thr' % tag]), g['int_%s_keep' % tag]
        if route == 'fused' and len(d) > 1024:
            d = d[:1024]
            keep = oracle.hbb_nms(d, thr)
        assert np.array_equal(nms(t(d), thr)[1].cpu().numpy(), keep), tag
        assert np.array_equal(nms(t(d), thr, strict=True)[1].cpu().numpy(), oracle.hbb_nms(d, thr, strict=True)), tag
        grp = (np.arange(len(d)) % 5).astype(np.int32)
        assert np.array_equal(nms(t(d), thr, groups=grp)[1].cpu().numpy(), oracle.hbb_nms(d, thr, groups=grp)), tag


def test_hbb_nms_malformed_and_nonpositive_threshold():
    from aidet_amd import nms, rnms
    d = np.array([[0, 0, 10, 10, 0.9], [1, 1, 11, 11, 0.8], [20, 5, 3, 9, 0.7],      # x2 < x1 - 1: inert
                  [2, 2, 9, 9, np.nan], [np.nan, 0, 5, 5, 0.6], [0, 0, 10, 10, 0.5],
                  [4, 4, 3, 3, 0.4]], np.float32)                                    # w = h = 0: inert
    assert np.array_equal(nms(t(d), 0.5)[1].cpu().numpy(), oracle.hbb_nms(d, 0.5))
    # thr <= 0 under `>=`: what nms_cpu.cpp:56 returns (round 5; pinned by hbb_nms_nonpos.npz in test_round5_gpu.py)
    for thr in (0.0, -1.0):
        assert np.array_equal(nms(t(d), thr)[1].cpu().numpy(), oracle.hbb_nms(d, thr))
    nms(t(d), 0.0, strict=True)            # `>` with thr 0 is an ordinary NMS
    # rotated kinds: the AABB prefilter is part of the operator, every threshold is defined
    r = oracle.synth_boxes(600, 200, 77)
    for thr in (0.0, -0.5):
        assert np.array_equal(rnms(t(r), thr)[1].cpu().numpy(), oracle.rnms(r, thr))


def test_rotated_strict_compare(nms_path):
    from aidet_amd import rnms
    two = np.array([[10, 10, 8, 4, 0.3, 0.9], [10, 10, 8, 4, 0.3, 0.8]], np.float32)
    assert rnms(two, 1.0)[1].tolist() == [0] and rnms(two, 1.0, strict=True)[1].tolist() == [0, 1]
    d = oracle.synth_boxes(5000, 300, 4)
    d[1000:1500] = d[:500]                  # exact duplicates: IoU == 1
    d[1000:1500, 5] *= 0.5
    assert np.array_equal(rnms(t(d), 1.0)[1].cpu().numpy(), oracle.rnms(d, 1.0))
    assert np.array_equal(rnms(t(d), 1.0, strict=True)[1].cpu().numpy(), oracle.rnms(d, 1.0, strict=True))
    assert len(oracle.rnms(d, 1.0)) < len(oracle.rnms(d, 1.0, strict=True))


def test_soft_nms_matches_reference(golden):
    """soft_nms_cpu_kernel (nms_cpu.cpp:71-210) through nms_wrapper.soft_nms (:63-118): the
    fixtures were produced by the reference's own build."""
    from aidet_amd.nms_wrapper import soft_nms
    g = golden('soft_nms.npz')
    nd, inds = soft_nms(g['doc6_dets'], 0.7, sigma=0.5)
    assert isinstance(nd, np.ndarray) and inds.dtype == np.int64 and len(inds) == 3
    assert np.array_equal(inds, g['doc6_inds']) and np.array_equal(nd, g['doc6_new_dets'])
    nd, inds = soft_nms(torch.from_numpy(g['test4_dets']), 0.7)       # tests/test_soft_nms.py: 4 outputs
    assert isinstance(nd, torch.Tensor) and not nd.is_cuda and inds.dtype == torch.long
    assert np.array_equal(inds.numpy(), g['test4_inds']) and np.array_equal(nd.numpy(), g['test4_new_dets'])
    for tag in 'abcde':
        thr, sigma, min_score = (float(v) for v in g['rand_%s_cfg' % tag])
        method = str(g['rand_%s_method' % tag])
        nd, inds = soft_nms(t(g['rand_%s_dets' % tag]), thr, method, sigma, min_score)
        assert nd.is_cuda and inds.is_cuda
        assert np.array_equal(inds.cpu().numpy(), g['rand_%s_inds' % tag]), tag
        want = g['rand_%s_new_dets' % tag]
        assert np.array_equal(nd.cpu().numpy()[:, :4], want[:, :4]), tag
        # (gaussian too: the kernels restate glibc's expf -- obb_softnms.hip soft_exp -- so the scores are the reference's bits)
        assert np.array_equal(nd.cpu().numpy()[:, 4], want[:, 4]), tag
    d64 = g['doc6_dets'].astype(np.float64)
    nd, inds = soft_nms(d64, 0.7, sigma=0.5)
    assert nd.dtype == np.float64 and inds.tolist() == g['doc6_inds'].tolist()
    with pytest.raises(ValueError):
        soft_nms(g['doc6_dets'], 0.7, method='bogus')
    with pytest.raises(TypeError):
        soft_nms([[0, 0, 1, 1, 1]], 0.7)
    nd, inds = soft_nms(np.zeros((0, 5), np.float32), 0.5)
    assert nd.shape == (0, 5) and inds.shape == (0,)


@pytest.mark.parametrize('n,canvas,method', [(1, 10, 'linear'), (2049, 400, 'linear'), (5000, 500, 'gaussian'),
                                             (3000, 120, 'linear')])
def test_soft_nms_vs_oracle_sizes(n, canvas, method):
    """n > 2048 keeps the state in the workspace instead of LDS; dense scenes drop a lot."""
    from aidet_amd.nms_wrapper import soft_nms
    rng = np.random.default_rng(n)
    x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
    d = np.stack([x1, y1, x1 + rng.uniform(4, 60, n), y1 + rng.uniform(4, 60, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    d[::4, 4] = np.float32(0.25)
    nd, inds = soft_nms(t(d), 0.3, method, 0.5, 0.05)
    rd, ri = oracle.soft_nms(d, 0.3, method, 0.5, 0.05)
    assert np.array_equal(inds.cpu().numpy(), ri)
    assert np.abs(nd.cpu().numpy() - rd).max() <= (0.0 if method == 'linear' else 1e-6)


def test_nms_async_tickets_match_sync_calls():
    """obb_nms_async_f32: several calls queued on one stream, one synchronisation."""
    from aidet_amd import nms, nms_wrapper, rnms
    sets, want = [], []
    for i, n in enumerate((0, 1, 100, 2000, 2000, 30000)):
        d = oracle.synth_boxes(n, 40.0 * max(n, 1) ** 0.5, 50 + i)
        sets.append(('rnms', t(d), 0.4 + 0.05 * i))
        want.append(oracle.rnms(d, 0.4 + 0.05 * i))
    rng = np.random.default_rng(1)
    x1, y1 = rng.uniform(0, 300, 2000), rng.uniform(0, 300, 2000)
    h = np.stack([x1, y1, x1 + rng.uniform(5, 50, 2000), y1 + rng.uniform(5, 50, 2000),
                  (rng.permutation(2000) + .5) / 2000], 1).astype(np.float32)
    sets.append(('nms', t(h), 0.7))
    want.append(oracle.hbb_nms(h, 0.7))
    tickets = [nms_wrapper.nms_enqueue(d, thr, op=op) for op, d, thr in sets]
    keeps = nms_wrapper.nms_collect(tickets)
    for k, w in zip(keeps, want):
        assert k.dtype == torch.int64 and np.array_equal(k.cpu().numpy(), w)
    # overflow of the default pair workspace inside a ticket -> repeated synchronously
    dup = np.tile(oracle.synth_boxes(1, 10, 3), (3000, 1))
    dup[:, 5] = (np.arange(3000, 0, -1) / 3001).astype(np.float32)
    tk = nms_wrapper.nms_enqueue(t(dup), 0.5, op='rnms')
    assert nms_wrapper.nms_collect([tk])[0].tolist() == [0]


def test_rnms_dense_all_overlap_needs_the_retry_loop():
    """20 000 mutually overlapping boxes = 2e8 pairs: the default workspace (64 n pairs)
    overflows, the wrapper loops on OBB_EWORKSPACE with the capacity the library reports."""
    from aidet_amd import rnms
    n = 20000
    rng = np.random.default_rng(8)
    d = np.zeros((n, 6), np.float32)
    d[:, 0] = 500 + rng.uniform(-2, 2, n); d[:, 1] = 500 + rng.uniform(-2, 2, n)
    d[:, 2] = 200; d[:, 3] = 100; d[:, 4] = rng.uniform(-0.02, 0.02, n)
    d[:, 5] = (rng.permutation(n) + 0.5) / n
    kept, inds = rnms(t(d), 0.5)
    assert inds.tolist() == [int(np.argmax(d[:, 5]))]
    kept, inds = rnms(t(d), 0.999)       # near-duplicates at a threshold almost nothing reaches
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.999))


def test_multiclass_nms_reference_golden_on_gpu(golden):
    """bbox_nms.py:6-76 captured outputs (mc_bboxes / mc_labels) through the HIP nms."""
    from aidet_amd.post_processing import multiclass_nms
    g = golden('multiclass.npz')
    cfg = dict(type='nms', iou_thr=float(g['iou_thr']))
    for grouped in (False, True):
        det, lab = multiclass_nms(t(g['multi_bboxes']), t(g['multi_scores']), float(g['score_thr']), cfg,
                                  int(g['max_num']), grouped=grouped)
        assert np.array_equal(det.cpu().numpy(), g['mc_bboxes']) and np.array_equal(lab.cpu().numpy(), g['mc_labels'])
    from aidet_amd.post_processing import multiclass_nms_with_index
    w = golden('multiclass_soft.npz')        # the same inputs through type='soft_nms' in the reference
    soft = dict(type='soft_nms', iou_thr=0.5, min_score=0.05)
    det, lab = multiclass_nms(t(g['multi_bboxes']), t(g['multi_scores']), float(g['score_thr']), soft, 100)
    assert np.array_equal(det.cpu().numpy(), w['mc_bboxes']) and np.array_equal(lab.cpu().numpy(), w['mc_labels'])
    b, l, ci, ki = multiclass_nms_with_index(t(g['multi_bboxes']), t(g['multi_scores']), float(g['score_thr']), soft, 100)
    assert np.array_equal(b.cpu().numpy(), w['idx_bboxes']) and np.array_equal(l.cpu().numpy(), w['idx_labels'])
    assert all(np.array_equal(k.cpu().numpy(), w['idx_keep_%d' % i]) for i, k in enumerate(ki))


def test_tiled_inference_merge_hip_equals_oracle_ops():
    """configs[4] on one GPU with the HIP rnms vs the same function with the oracle's op."""
    from aidet_amd import dota_io, dota_merge
    from test_dist_merge import _tiled_inputs, oracle_nms, NUM_CLASSES, CLASS_THRS
    args, _ = _tiled_inputs(0, 1)
    kept, keep, allrows = dota_merge.tiled_inference_merge(*[a.cuda() for a in args], NUM_CLASSES, 0.5, 0.3, CLASS_THRS)
    rk, rkeep, rall = dota_merge.tiled_inference_merge(*args, NUM_CLASSES, 0.5, 0.3, CLASS_THRS, nms_fn=oracle_nms)
    assert np.array_equal(allrows.cpu().numpy(), rall.numpy())
    assert np.array_equal(keep.cpu().numpy(), rkeep.numpy())


def test_bench_distributed_path_runs_on_one_gpu():
    """`bench.py` N > 1 code path (RCCL process group, all_gather_into_tensor, merged NMS, the
    single-rank cross-check) forced onto one GPU: what an 8-GPU SCALE run would time."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OBB_BENCH_FORCE_DIST='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    for layout in ('replicated', 'routed'):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                            '--no-cpu', '--no-extras', '--no-cfg5-large', '--merge-layout', layout],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        j = json.loads(line)
        # every line carries both legs: the replica leg is `value` (weak), the cfg5 leg sits in `config` as scalars
        assert j['scaling'] == 'weak' and j['rccl_ranks'] == 1 and j['n_gpus'] == 1 and j['value'] > 1e7
        c = j['config']
        assert c['cfg5_layout'] == layout and c['cfg5_keep_ok'] is True and c['cfg5_boxes_per_s'] > 1e7
        assert c['iou_pairs_per_s'] > 1e10 and c['cfg5_speedup_vs_single_rank'] > 0
        d = j['cfg5'][layout]
        assert d['keep_equals_single_rank_run'] is True and d['same_keep_on_all_ranks'] is True
        assert all(not isinstance(v, (dict, list)) for v in list(j['roofline'].values()) + list(c.values()))


def test_pyramid_roi_align_one_launch_equals_per_level_calls():
    """SingleRoIExtractor (single_level.py:89-107): the inference path pools all levels in ONE
    launch (obb_roi_align_levels_forward_f32); it must equal the per-level loop (taken when a
    gradient is needed) bit for bit, and the oracle's V1 restatement level by level."""
    from aidet_amd.roi_layers import RotatedSingleRoIExtractor, SingleRoIExtractor
    rng = np.random.default_rng(12)
    feats = [rng.standard_normal((2, 40, 64 >> i, 80 >> i)).astype(np.float32) for i in range(4)]
    K = 300
    x1, y1 = rng.uniform(0, 250, K), rng.uniform(0, 200, K)
    size = np.exp(rng.uniform(np.log(6), np.log(600), (K, 2)))
    rois = np.concatenate([rng.integers(0, 2, (K, 1)), x1[:, None], y1[:, None], x1[:, None] + size[:, :1],
                           y1[:, None] + size[:, 1:]], 1).astype(np.float32)
    ext = SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=2), 40, [4, 8, 16, 32])
    ft = [t(f) for f in feats]
    with torch.no_grad():
        fused = ext(ft, t(rois))
    looped = ext([f.clone().requires_grad_(True) for f in ft], t(rois))
    assert torch.equal(fused, looped.detach())
    lv = ext.map_roi_levels(torch.from_numpy(rois), 4).numpy()
    assert len(set(lv.tolist())) == 4
    for l in range(4):
        ref = oracle.roi_align_forward(feats[l], rois[lv == l], 7, 1.0 / (4 << l), 2, version=1)
        assert np.array_equal(fused.cpu().numpy()[lv == l], ref), l
    # rotated extractor
    rr = np.concatenate([rois[:, :1], oracle.synth_boxes(K, 300, 4)[:, :5]], 1).astype(np.float32)
    rr[:, 3:5] *= rng.choice([0.5, 1.0, 4.0], (K, 1))
    rext = RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=(5, 7), sample_num=2), 40, [4, 8, 16, 32])
    with torch.no_grad():
        fused = rext(ft, t(rr))
    looped = rext([f.clone().requires_grad_(True) for f in ft], t(rr))
    assert torch.equal(fused, looped.detach())
    # a level index outside the pyramid gives zeros, like a batch index outside the batch
    from aidet_amd.roi_layers import pyramid_roi_align
    bad = pyramid_roi_align(ft, t(rois[:5]), torch.tensor([0, 9, -1, 3, 1]), 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, 1)
    assert float(bad[1].abs().max()) == 0.0 and float(bad[2].abs().max()) == 0.0 and float(bad[0].abs().max()) > 0


def test_horizontal_bbox_overlaps_matches_reference_bit_for_bit(golden):
    """aidet_amd.bbox_overlaps = the reference's bbox_overlaps (geometry.py:4-88) on the GPU:
    bit-equal to the values the reference produced (hbb_iou.npz), and to the oracle on
    degenerate rows (negative / zero extents, NaN, inf) where the reference yields -0 / NaN."""
    from aidet_amd import bbox_overlaps
    g = golden('hbb_iou.npz')
    eq = lambda x, y: np.array_equal(x, y, equal_nan=True)
    assert eq(bbox_overlaps(t(g['doc_b1']), t(g['doc_b2'])).cpu().numpy(), g['doc_iou'])
    assert eq(bbox_overlaps(t(g['rand_b1']), t(g['rand_b2'])).cpu().numpy(), g['rand_iou'])
    assert eq(bbox_overlaps(t(g['rand_b1']), t(g['rand_b2']), 'iof').cpu().numpy(), g['rand_iof'])
    assert eq(bbox_overlaps(t(g['rand_b1']), t(g['rand_b2'][:200]), is_aligned=True).cpu().numpy(), g['rand_iou_aligned'])
    assert eq(bbox_overlaps(t(g['rand_b1']), t(g['rand_b2'][:200]), 'iof', True).cpu().numpy(), g['rand_iof_aligned'])
    assert eq(bbox_overlaps(t(g['int_b1']), t(g['int_b2'])).cpu().numpy(), g['int_iou'])
    assert tuple(bbox_overlaps(torch.zeros(0, 4), t(g['doc_b2'])).shape) == (0, 3)          # geometry.py:37-44
    assert tuple(bbox_overlaps(t(g['doc_b1']), torch.zeros(0, 4), is_aligned=False).shape) == (3, 0)
    rng = np.random.default_rng(9)
    for m, n, canvas in ((700, 1300, 300), (2500, 4100, 4000), (33, 5000, 60)):
        x1, y1 = rng.uniform(0, canvas, m), rng.uniform(0, canvas, m)
        a = np.stack([x1, y1, x1 + rng.uniform(2, 90, m), y1 + rng.uniform(2, 90, m)], 1).astype(np.float32)
        x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
        b = np.stack([x1, y1, x1 + rng.uniform(2, 90, n), y1 + rng.uniform(2, 90, n)], 1).astype(np.float32)
        a[3] = [50, 50, 40, 90]; a[4] = [10, 10, 9, 9]; a[5] = [np.nan, 0, 5, 5]; a[6] = [0, 0, np.inf, 5]
        b[7] = [60, 60, 20, 20]; b[8] = [10, 10, 9, 9]; b[9] = [0, np.nan, 5, 5]
        for mode in ('iou', 'iof'):
            got = bbox_overlaps(t(a), t(b), mode).cpu().numpy()
            assert eq(got, oracle.bbox_overlaps(a, b, mode)), (m, n, mode)
    # same boxes as theta = 0 OBBs with the +1 widths: the rotated op agrees within 1e-5 where both are defined
    from aidet_amd import rbbox_overlaps
    rb = lambda h: t(hbb_to_rbox(h))
    reg = bbox_overlaps(t(g['rand_b1']), t(g['rand_b2'])).cpu().numpy()
    assert np.abs(rbbox_overlaps(rb(g['rand_b1']), rb(g['rand_b2'])).cpu().numpy() - reg).max() <= 1e-5


def test_horizontal_max_iou_assigner_equals_reference(golden):
    """MaxIoUAssigner on horizontal boxes with the HIP bbox_overlaps: the reference's assignment
    (assigner.npz), max_overlaps bit for bit."""
    from aidet_amd.assigners import MaxIoUAssigner
    g = golden('assigner.npz')
    boxes, gts, ign, labels = t(g['boxes']), t(g['gts']), t(g['ign']), t(g['labels'])
    cases = (('a', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3), False),
             ('b', dict(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.0, gt_max_assign_all=False), False),
             ('c', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=0.5), True))
    for tag, kw, use_ign in cases:
        r = MaxIoUAssigner(**kw).assign(boxes, gts, ign if use_ign else None, labels)
        assert np.array_equal(r.gt_inds.cpu().numpy(), g['assign_%s_gt_inds' % tag]), tag
        assert np.array_equal(r.labels.cpu().numpy(), g['assign_%s_labels' % tag]), tag
        assert np.array_equal(r.max_overlaps.cpu().numpy(), g['assign_%s_max_overlaps' % tag]), tag


def test_merge_row_kernels_equal_the_tensor_expressions():
    """obb_merge.hip (patch groups, rows to image coordinates, row split, row gather) against the torch
    composition of aidet_amd/dota_merge.py evaluated on the CPU: bit-identical, and the merged keep of
    every row layout equals the one obtained through the generic wrappers."""
    from aidet_amd import dota_merge, nms_wrapper
    rng = np.random.default_rng(21)
    n, P, C = 5000, 37, 15
    dets = oracle.synth_boxes(n, 150, 4)
    labels = torch.from_numpy(rng.integers(0, C, n))
    pidx = torch.from_numpy(rng.integers(0, P, n))
    pxy = torch.from_numpy(rng.integers(0, 8000, (P, 2)).astype(np.float32))
    pim = torch.from_numpy(rng.integers(0, 5, P))
    keep = torch.from_numpy(np.sort(rng.choice(n, 3000, replace=False)))
    d = torch.from_numpy(dets)
    for rate in (1.0, 0.5, 1.5):
        for kp in (None, keep):
            ref = dota_merge.patch_rows_to_image(d, labels, pidx, pxy, pim, rate, keep=kp)
            got = dota_merge.patch_rows_to_image(d.cuda(), labels.cuda(), pidx.cuda(), pxy.cuda(), pim.cuda(), rate,
                                                 keep=None if kp is None else kp.cuda())
            assert np.array_equal(got.cpu().numpy(), ref.numpy()), (rate, kp is None)
    # per-patch NMS: fused group ids vs the int64 expression through the wrapper
    k1 = dota_merge.local_patch_nms(d.cuda(), labels.cuda(), pidx.cuda(), 0.5, C)
    k2 = dota_merge.local_patch_nms(d.cuda(), labels.cuda(), pidx.cuda(), 0.5, C, nms_fn=nms_wrapper.rnms)
    assert torch.equal(k1, k2) and 0 < k1.numel() < n
    # merged NMS: every row layout, flat / per-class thresholds, strict
    rows8 = dota_merge.patch_rows_to_image(d, labels, pidx, pxy * 0.05, pim, 1.0)
    poly = np.concatenate([oracle.rbox_to_poly(rows8[:, :5].numpy()).astype(np.float32), rows8[:, 5:].numpy()], 1)
    hb = rows8[:, :5].numpy()
    box = np.stack([hb[:, 0] - hb[:, 2] / 2, hb[:, 1] - hb[:, 3] / 2, hb[:, 0] + hb[:, 2] / 2, hb[:, 1] + hb[:, 3] / 2], 1)
    rows7 = np.concatenate([box, rows8[:, 5:].numpy()], 1).astype(np.float32)
    thr_tab = [0.1 + 0.03 * c for c in range(C)]
    for rows, op in ((rows8.numpy(), nms_wrapper.rnms), (poly, nms_wrapper.poly_nms), (rows7, nms_wrapper.nms)):
        r = torch.from_numpy(rows).cuda()
        for cthr in (None, thr_tab):
            for strict in (False, True):
                kept, kp = dota_merge.merge_detections(r, 0.3, C, cthr, strict=strict)
                fn = (lambda dd, tt, groups=None, thrs=None, **kw: op(dd, tt, groups=groups, thrs=thrs, **kw))
                rkept, rkp = dota_merge.merge_detections(r, 0.3, C, cthr, nms_fn=fn, strict=strict)
                assert torch.equal(kp, rkp) and torch.equal(kept, rkept), (rows.shape[1], cthr is None, strict)
                assert 0 < kp.numel() < len(rows)
    # the sharded layout: the groups of one rank only; the union over the ranks is the whole keep
    r8 = rows8.cuda()
    full = dota_merge.merge_detections(r8, 0.3, C, thr_tab)[1]
    parts = []
    for rank in range(3):
        kept, kp = dota_merge.merge_detections(r8, 0.3, C, thr_tab, only_groups=(3, rank))
        fn = (lambda dd, tt, groups=None, thrs=None, **kw: nms_wrapper.rnms(dd, tt, groups=groups, thrs=thrs, **kw))
        rkept, rkp = dota_merge.merge_detections(r8, 0.3, C, thr_tab, nms_fn=fn, only_groups=(3, rank))
        assert torch.equal(kp, rkp) and torch.equal(kept, rkept) and kp.numel() > 0
        parts.append(kp)
    assert torch.equal(torch.cat(parts).sort()[0], full)
    # gather + degenerate inputs
    idx = torch.from_numpy(rng.integers(0, n, 777)).cuda()
    assert torch.equal(dota_merge.gather_rows(rows8.cuda(), idx), rows8.cuda()[idx])
    assert dota_merge.gather_rows(rows8.cuda(), idx[:0]).shape == (0, 8)
    e = dota_merge.patch_rows_to_image(d.cuda(), labels.cuda(), pidx.cuda(), pxy.cuda(), pim.cuda(), 1.0, keep=keep[:0].cuda())
    assert e.shape == (0, 8)
    bad = pidx.clone(); bad[0] = P + 3     # a patch index out of range: zero row, label = image = -1
    z = dota_merge.patch_rows_to_image(d.cuda(), labels.cuda(), bad.cuda(), pxy.cuda(), pim.cuda())
    assert z[0].tolist() == [0, 0, 0, 0, 0, 0, -1, -1] and torch.equal(z[1:].cpu(), dota_merge.patch_rows_to_image(d, labels, pidx, pxy, pim)[1:])


def test_grouped_rnms_large_call_matches_oracle_group_by_group():
    """4e5 boxes in 20 000 groups sharing one coordinate range (the merged NMS of tiled inference at
    a size where the sort uses 2^18 cells, the cull kernel its supertiles and the compaction four
    slices per block): groups never interact, so the oracle can check any of them on its own."""
    from aidet_amd import rnms
    rng = np.random.default_rng(31)
    n, ng = 400000, 20000
    d = oracle.synth_boxes(n, 250.0, 17)                # ~20 boxes per group on a 250 px canvas: dense
    groups = rng.integers(0, ng, n).astype(np.int32) * 7919 - 50000000     # arbitrary ids
    thrs = rng.choice([0.1, 0.3, 0.5], n).astype(np.float32)
    for kw in ({}, {'thrs': thrs}):
        keep = rnms(t(d), 0.4, groups=groups, **kw)[1].cpu().numpy()
        assert np.all(np.diff(keep) > 0) and 0.3 * n < len(keep) < n
        kept = np.zeros(n, bool); kept[keep] = True
        for g in np.unique(groups)[rng.integers(0, ng - 1, 150)]:
            idx = np.nonzero(groups == g)[0]
            ref = oracle.rnms(d[idx], 0.4, thrs=None if not kw else thrs[idx])
            assert np.array_equal(np.nonzero(kept[idx])[0], ref), int(g)


@pytest.mark.parametrize('n,canvas,ng', [(30000, 2500.0, 0), (70000, 3000.0, 900), (150000, 9000.0, 0),
                                         (300000, 6000.0, 40000), (600000, 30000.0, 7)])
def test_rnms_routes_agree_at_sizes_the_oracle_cannot_reach(n, canvas, ng, nms_route):
    """Every route computes the same greedy fixpoint: direct (n <= 65 536), sorted with the clip in
    line, sorted with candidate lists -- dense scenes, few / very many groups, per-box thresholds."""
    from aidet_amd import rnms
    rng = np.random.default_rng(n)
    d = t(oracle.synth_boxes(n, canvas, n % 97))
    groups = None if ng == 0 else torch.from_numpy(rng.integers(0, ng, n).astype(np.int32)).cuda()
    thrs = torch.from_numpy(rng.choice([0.2, 0.45, 0.7], n).astype(np.float32)).cuda() if n % 20000 == 0 else None
    ref = None
    for route in ('listed', 'sorted', 'direct', 'auto'):
        if route == 'direct' and n > 65536:
            continue
        nms_route(route)
        keep = rnms(d, 0.5, groups=groups, thrs=thrs)[1]
        if ref is None:
            ref = keep
            assert 0.2 * n < keep.numel() < n
        assert torch.equal(keep, ref), route


def test_get_det_rbboxes_pointobb_and_hobb_encodings():
    """RBBoxHead's other two encodings (rbbox_head.py:264-269): 'pointobb' decodes to corner points and
    gets the true polygon NMS per class (or the keep re-use), 'hobb' the keep re-use only."""
    from aidet_amd.post_processing import get_det_rbboxes, multiclass_nms_with_index
    from aidet_amd.rbbox_transforms import delta2hobb, delta2pointobb
    rng = np.random.default_rng(8)
    n, ncls = 260, 5
    x1 = rng.uniform(0, 300, n); y1 = rng.uniform(0, 300, n)
    rois = np.stack([np.zeros(n), x1, y1, x1 + rng.uniform(12, 80, n), y1 + rng.uniform(12, 80, n)], 1).astype(np.float32)
    cls_score = (rng.standard_normal((n, ncls)) * 2).astype(np.float32)
    m8, s8 = (0.,) * 8, (.1, .1, .2, .2, .1, .1, .2, .2)
    pred8 = (rng.standard_normal((n, ncls * 8)) * 0.3).astype(np.float32)
    cfg = dict(score_thr=0.05, max_per_img=80, polygon_nms_iou_thr=0.3, encode='pointobb')
    det, lab = get_det_rbboxes(t(rois), t(cls_score), t(pred8), (512, 512, 3), 1.0, cfg=cfg, target_means=m8, target_stds=s8)
    det, lab = det.cpu().numpy(), lab.cpu().numpy()
    assert det.shape[1] == 9 and len(det) == len(lab) <= 80
    # oracle: decode on the CPU, per-class polygon NMS, best 80
    sc = torch.softmax(torch.from_numpy(cls_score), 1).numpy()
    quads = delta2pointobb(torch.from_numpy(rois[:, 1:]), torch.from_numpy(pred8), m8, s8).numpy()
    rows, labs = [], []
    for c in range(1, ncls):
        m = sc[:, c] > 0.05
        d = np.concatenate([quads[m, c * 8:(c + 1) * 8], sc[m, c:c + 1]], 1).astype(np.float32)
        k = oracle.poly_nms(d, 0.3)
        rows.append(d[k]); labs.append(np.full(len(k), c - 1))
    rows, labs = np.concatenate(rows), np.concatenate(labs)
    top = np.argsort(-rows[:, 8], kind='stable')[:80]
    assert np.abs(det[:, 8] - rows[top, 8]).max() <= 1e-6 and np.array_equal(lab, labs[top])
    assert np.abs(det[:, :8] - rows[top, :8]).max() <= 1e-3
    # keep re-use with both encodings: the rows the horizontal NMS kept, in its order
    bb = np.tile(rois[:, 1:], (1, ncls)).astype(np.float32)
    nms_cfg = dict(type='nms', iou_thr=0.5)
    for enc, pred, means, stds, dim, dec in (('pointobb', pred8, m8, s8, 8, delta2pointobb),
                                             ('hobb', (rng.standard_normal((n, ncls * 5)) * 0.3).astype(np.float32),
                                              (0.,) * 5, (.1, .1, .2, .2, .1), 5, delta2hobb)):
        _, hl, ci, ki = multiclass_nms_with_index(t(bb), t(cls_score).softmax(1), 0.05, nms_cfg, 80)
        d2, l2 = get_det_rbboxes(t(rois), t(cls_score), t(pred), (512, 512, 3), 2.0, rescale=True,
                                 cfg=dict(cfg, encode=enc), target_means=means, target_stds=stds,
                                 bbox_cls_inds=ci, bbox_keep_inds=ki, rotated_nms=False)
        assert d2.shape[1] == dim + 1 and torch.equal(l2, hl)
        full = dec(torch.from_numpy(rois[:, 1:]), torch.from_numpy(pred), means, stds).numpy() / 2.0
        got = d2.cpu().numpy()
        # every returned box is the decoded (and rescaled) box of its class for some row with that score
        for r, c in zip(got, l2.cpu().numpy()):
            cand = full[:, (c + 1) * dim:(c + 2) * dim]
            assert np.abs(cand - r[:dim]).max(1).min() <= 1e-3
    # 'hobb' with the true NMS: decided on the quadrilateral of every hobb, hobb rows returned
    from aidet_amd.rbbox_transforms import hobb_corner_points
    m5, s5 = (0.,) * 5, (.1, .1, .2, .2, .1)
    dh, lh = get_det_rbboxes(t(rois), t(cls_score), t(pred), (512, 512, 3), 1.0, cfg=dict(cfg, encode='hobb'),
                             target_means=m5, target_stds=s5)
    dh, lh = dh.cpu().numpy(), lh.cpu().numpy()
    assert dh.shape[1] == 6 and len(dh) == len(lh) <= 80
    hob = delta2hobb(torch.from_numpy(rois[:, 1:]), torch.from_numpy(pred), m5, s5)
    rows, labs = [], []
    for c in range(1, ncls):
        m = sc[:, c] > 0.05
        hc = hob[:, c * 5:(c + 1) * 5][torch.from_numpy(m)]
        quad = hobb_corner_points(hc).numpy().astype(np.float32)
        k = oracle.poly_nms(np.concatenate([quad, sc[m, c:c + 1]], 1).astype(np.float32), 0.3)
        rows.append(np.concatenate([hc.numpy()[k], sc[m, c:c + 1][k]], 1)); labs.append(np.full(len(k), c - 1))
    rows, labs = np.concatenate(rows), np.concatenate(labs)
    top = np.argsort(-rows[:, 5], kind='stable')[:80]
    assert np.array_equal(lh, labs[top]) and np.abs(dh - rows[top]).max() <= 1e-3
    with pytest.raises(ValueError):
        get_det_rbboxes(t(rois), t(cls_score), t(pred), (512, 512, 3), 1.0, cfg=dict(cfg, encode='maskobb'))


def test_bbox_overlaps_is_differentiable_like_the_reference_expression():
    """iou_loss.py:25 differentiates through bbox_overlaps(pred, target, is_aligned=True): values from the
    kernel, gradients equal to autograd of geometry.py's tensor expressions; the rotated op has its own gradient
    (tests/test_round3_gpu.py), the polygon op refuses."""
    from aidet_amd import bbox_overlaps, rbbox_overlaps
    from aidet_amd.rbbox_geometry import _hbb_overlaps_torch, pointobb_overlaps
    rng = np.random.default_rng(4)
    x1, y1 = rng.uniform(0, 200, 300), rng.uniform(0, 200, 300)
    a = np.stack([x1, y1, x1 + rng.uniform(5, 80, 300), y1 + rng.uniform(5, 80, 300)], 1).astype(np.float32)
    b = (a + rng.normal(0, 6, a.shape)).astype(np.float32)
    for aligned, mode in ((True, 'iou'), (False, 'iou'), (True, 'iof'), (False, 'iof')):
        p, q = t(a).requires_grad_(True), t(b[:64] if not aligned else b).requires_grad_(True)
        out = bbox_overlaps(p, q, mode, aligned)
        loss = (1 - out).pow(2).sum()
        loss.backward()
        pr, qr = t(a).requires_grad_(True), t(b[:64] if not aligned else b).requires_grad_(True)
        ref = _hbb_overlaps_torch(pr, qr, mode, aligned)
        (1 - ref).pow(2).sum().backward()
        assert float((out.detach() - ref.detach()).abs().max()) <= 1e-6
        assert float((p.grad - pr.grad).abs().max()) <= 1e-5 and float((q.grad - qr.grad).abs().max()) <= 1e-5
        assert float(p.grad.abs().sum()) > 0
    # only one side needs a gradient (the loss's target is detached)
    p = t(a).requires_grad_(True)
    bbox_overlaps(p, t(b), 'iou', True).sum().backward()
    assert p.grad is not None
    with pytest.raises(RuntimeError):
        pointobb_overlaps(torch.rand(4, 8, device='cuda', requires_grad=True), torch.rand(4, 8, device='cuda'))
    with torch.no_grad():      # target assignment: fine without a graph
        assert rbbox_overlaps(torch.rand(4, 5, device='cuda', requires_grad=True), torch.rand(4, 5, device='cuda')).shape == (4, 4)


@pytest.mark.parametrize('mode', ['gather', 'atomics'])
def test_roi_backward_formulations_match_the_oracle(mode):
    """Both backward formulations (roi_layers.BACKWARD) on rotated and horizontal RoIs incl. RoIs too large for
    the plan (atomics fallback inside the gather entry point), RoIs off the map, an empty batch index and two
    images; the gather result of plannable RoIs is bit-reproducible and needs no zero-filled map."""
    from aidet_amd import RRoIAlign, RoIAlign, roi_layers
    rng = np.random.default_rng(77)
    old = roi_layers.BACKWARD
    roi_layers.BACKWARD = mode
    try:
        feat = rng.standard_normal((2, 24, 72, 88)).astype(np.float32)
        K = 90
        r = oracle.synth_boxes(K, 300, 9)[:, :5]
        r[:6, 2:4] *= 6.0                    # large: footprints beyond 2047 pixels at scale 0.5
        r[6:10, :2] += 2000                  # far off the map
        rois = np.concatenate([rng.integers(0, 2, (K, 1)).astype(np.float32), r], 1)
        rois[10, 0] = 5                      # batch index out of range: no gradient
        g = rng.standard_normal((K, 24, 7, 7)).astype(np.float32)
        for scale, sn in ((0.5, 2), (0.25, 0)):
            f = t(feat).requires_grad_(True)
            out = RRoIAlign(7, scale, sample_num=sn)(f, t(rois))
            out.backward(t(g))
            ref = oracle.rroi_align_backward(g, rois, feat.shape, scale, sn)
            assert float((f.grad - t(ref)).abs().max()) <= 1e-4 * max(1.0, float(np.abs(ref).max())), (mode, scale, sn)
        hb = np.stack([r[:, 0] - r[:, 2] / 2, r[:, 1] - r[:, 3] / 2, r[:, 0] + r[:, 2] / 2, r[:, 1] + r[:, 3] / 2], 1)
        hrois = np.concatenate([rois[:, :1], hb], 1).astype(np.float32)
        for aligned in (False, True):     # horizontal kinds: the two formulations against each other
            grads = {}
            for m2 in ('gather', 'atomics'):
                roi_layers.BACKWARD = m2
                f = t(feat).requires_grad_(True)
                RoIAlign(7, 0.5, sample_num=2, aligned=aligned)(f, t(hrois)).backward(t(g))
                grads[m2] = f.grad
            roi_layers.BACKWARD = mode
            scale_g = max(1.0, float(grads['atomics'].abs().max()))
            assert float((grads['gather'] - grads['atomics']).abs().max()) <= 1e-4 * scale_g, aligned
            assert float(grads['gather'].abs().sum()) > 0
        if mode == 'gather':
            small = rois[12:].copy()          # plannable RoIs only: two runs are bit-identical
            f1, f2 = t(feat).requires_grad_(True), t(feat).requires_grad_(True)
            lay = RRoIAlign(7, 0.5, sample_num=2)
            lay(f1, t(small)).backward(t(g[12:])); lay(f2, t(small)).backward(t(g[12:]))
            assert torch.equal(f1.grad, f2.grad)
    finally:
        roi_layers.BACKWARD = old
