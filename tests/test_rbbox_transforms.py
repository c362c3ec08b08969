"""CPU tests: theta-OBB codec vs values captured from the reference
(tests/golden/thetaobb_codec.npz, anchor A5) and host-side RoI-layer mappings."""
import numpy as np
import pytest
import torch

from aidet_amd import rbbox_transforms as T


def test_codec_matches_reference(golden):
    g = golden('thetaobb_codec.npz')
    out = T.delta2thetaobb(torch.from_numpy(g['small_rois']), torch.from_numpy(g['small_deltas']))
    assert np.allclose(out.numpy(), g['small_thetaobb'], atol=1e-6)
    # the printed values of SURVEY 8c A5
    assert np.allclose(out.numpy(), [[5.5, 13.5, 13.4986, 18.0967, -1.5208],
                                     [14.5, 9.5, 20, 10, -1.5708]], atol=1e-4)
    R, D, stds = torch.from_numpy(g['rois']), torch.from_numpy(g['deltas']), g['stds'].tolist()
    dec = T.delta2thetaobb(R, D, [0, 0, 0, 0, 0], stds)
    assert np.array_equal(dec.numpy(), g['thetaobb'])
    enc = T.thetaobb2delta(R, torch.from_numpy(g['gt']), [0, 0, 0, 0, 0], stds)
    assert np.allclose(enc.numpy(), g['enc'], atol=1e-6)
    # decode(encode(x)) round trip through the reference's +1 size convention
    back = T.delta2thetaobb(R, enc, [0, 0, 0, 0, 0], stds)
    gt = torch.from_numpy(g['gt'])
    assert np.allclose(back[:, :2].numpy(), gt[:, :2].numpy(), atol=1e-3)
    assert np.allclose(back[:, 4].numpy(), gt[:, 4].numpy(), atol=1e-5)


def test_flip_rescale_mapping_match_reference(golden):
    g = golden('thetaobb_codec.npz')
    assert np.array_equal(T.thetaobb_flip(g['flip_in'], (600, 800, 3)), g['flip_out'])
    assert np.array_equal(T.thetaobb_flip(torch.from_numpy(g['flip_in']), (600, 800, 3)).numpy(), g['flip_out'])
    x = torch.from_numpy(g['flip_in'].copy())
    assert np.array_equal(T.thetaobb_rescale(x, 1.7).numpy(), g['rescale_out'])
    x = torch.from_numpy(g['flip_in'].copy())
    assert np.array_equal(T.thetaobb_rescale(x, 1.7, reverse_flag=True).numpy(), g['rescale_rev_out'])
    # flip is an involution; mapping_back inverts mapping
    f2 = T.thetaobb_flip(T.thetaobb_flip(g['flip_in'], (600, 800, 3)), (600, 800, 3))
    assert np.allclose(f2, g['flip_in'], atol=1e-4)
    t = torch.from_numpy(g['flip_in'][:, :5].copy())
    m = T.thetaobb_mapping_back(T.thetaobb_mapping(t, (600, 800, 3), 1.0, True), (600, 800, 3), 1.0, True)
    assert np.allclose(m.numpy(), t.numpy(), atol=1e-4)


def test_rbbox2result():
    det = torch.arange(18, dtype=torch.float32).reshape(3, 6)
    res = T.rbbox2result(det, torch.tensor([1, 0, 1]), 4)
    assert [r.shape for r in res] == [(1, 6), (2, 6), (0, 6)]
    assert [r.shape for r in T.rbbox2result(det[:0], torch.zeros(0).long(), 4)] == [(0, 6)] * 3


def test_rotated_extractor_level_mapping_matches_reference_rule():
    """single_level.py:69-73: floor(log2(scale/56 + 1e-6)) clamped, scale = sqrt(w*h)."""
    from aidet_amd import RotatedSingleRoIExtractor
    ex = RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=7, sample_num=2), 256, [4, 8, 16, 32])
    wh = torch.tensor([10., 55.9, 111.9, 112.1, 224.1, 448.1, 2000.])
    rois = torch.stack([torch.zeros(7), wh, wh, wh, wh, torch.zeros(7)], 1)
    assert ex.map_roi_levels(rois, 4).tolist() == [0, 0, 0, 1, 2, 3, 3]
    r2 = ex.roi_rescale(rois, 1.5)
    assert torch.allclose(r2[:, 3], wh * 1.5) and torch.equal(r2[:, 1], rois[:, 1])
    assert ex.num_inputs == 4 and [l.spatial_scale for l in ex.roi_layers] == [0.25, 0.125, 0.0625, 0.03125]


def test_assign_wrt_overlaps_matches_reference(golden):
    """max_iou_assigner.py:114-202 on the reference's own overlap matrix."""
    from aidet_amd.assigners import assign_wrt_overlaps
    g = golden('assigner.npz')
    ov = torch.from_numpy(g['overlaps'])
    labels = torch.from_numpy(g['labels'])
    r = assign_wrt_overlaps(ov.clone(), labels, 0.5, 0.4, 0.3, True)
    assert np.array_equal(r.gt_inds.numpy(), g['assign_a_gt_inds'])
    assert np.array_equal(r.labels.numpy(), g['assign_a_labels'])
    assert np.array_equal(r.max_overlaps.numpy(), g['assign_a_max_overlaps'])
    r = assign_wrt_overlaps(ov.clone(), labels, 0.7, (0.1, 0.3), 0.0, False)
    assert np.array_equal(r.gt_inds.numpy(), g['assign_b_gt_inds'])
    assert np.array_equal(r.labels.numpy(), g['assign_b_labels'])
    # reference doctest (max_iou_assigner.py:81-87) and empty cases (:131-147)
    r = assign_wrt_overlaps(torch.tensor([[0.9, 0.0]]), None, 0.5, 0.5)
    assert r.gt_inds.tolist() == [1, 0]
    r = assign_wrt_overlaps(torch.zeros((0, 3)), torch.zeros(0).long())
    assert r.gt_inds.tolist() == [0, 0, 0] and r.labels.tolist() == [0, 0, 0]
    r = assign_wrt_overlaps(torch.zeros((2, 0)))
    assert r.gt_inds.numel() == 0


def test_pointobb_and_hobb_formats_reproduce_reference(golden):
    """The other two head encodings of core/rbbox/transforms.py and their host-side helpers, against
    values captured from the reference (tests/golden/obb_codecs.npz)."""
    from aidet_amd import rbbox_transforms as T
    g = golden('obb_codecs.npz')
    R, gt8, D8 = (torch.from_numpy(g[k]) for k in ('p_rois', 'p_gt', 'p_deltas'))
    m8, s8 = g['p_means'].tolist(), g['p_stds'].tolist()
    assert np.array_equal(T.pointobb2delta(R, gt8, m8, s8).numpy(), g['p_enc'])
    assert np.array_equal(T.pointobb2delta(R, gt8).numpy(), g['p_enc_default'])
    assert np.array_equal(T.delta2pointobb(R, D8, m8, s8).numpy(), g['p_dec'])
    assert np.array_equal(T.delta2pointobb(R, D8[:, :8]).numpy(), g['p_dec_default'])
    gt5, D5 = torch.from_numpy(g['h_gt']), torch.from_numpy(g['h_deltas'])
    m5, s5 = g['h_means'].tolist(), g['h_stds'].tolist()
    assert np.array_equal(T.hobb2delta(R, gt5, m5, s5).numpy(), g['h_enc'])
    assert np.array_equal(T.delta2hobb(R, D5, m5, s5).numpy(), g['h_dec'])
    assert np.array_equal(T.delta2hobb(R, D5[:, :5]).numpy(), g['h_dec_default'])
    # round trip: decode(encode(gt)) == gt up to rounding
    back = T.delta2pointobb(R, T.pointobb2delta(R, gt8, m8, s8), m8, s8)
    assert float((back - gt8).abs().max()) < 1e-3
    # rescale works in place and returns its argument
    x = gt8.clone(); y = T.pointobb_rescale(x, 1.3)
    assert y is x and np.array_equal(x.numpy(), g['p_rescale'])
    assert np.array_equal(T.pointobb_rescale(gt8.clone(), 1.3, reverse_flag=True).numpy(), g['p_rescale_rev'])
    assert np.array_equal(T.hobb_rescale(gt5.clone(), 0.8).numpy(), g['h_rescale'])
    assert np.array_equal(T.hobb_rescale(gt5.clone(), 0.8, reverse_flag=True).numpy(), g['h_rescale_rev'])
    # point-order helpers, flip and mappings (host side, numpy)
    q = g['q_in']
    assert np.array_equal(np.array([T.pointobb2bbox(r.tolist()) for r in q]), g['q_bbox'])
    assert np.array_equal(np.array([T.pointobb_extreme_sort(r.tolist()) for r in q]), g['q_extreme'])
    assert np.array_equal(np.array([T.pointobb_best_point_sort(r.tolist()) for r in q]), g['q_best'])
    assert np.array_equal(T.pointobb_flip(q, (600, 800, 3)), g['q_flip'])
    one = T.pointobb_flip(q[3], (600, 800, 3))
    assert one.shape == (8,) and np.array_equal(one, g['q_flip_one'])
    assert np.array_equal(T.pointobb_mapping(q, (900, 1200, 3), 1.5, True), g['q_map'])
    assert np.array_equal(T.pointobb_mapping(q, (900, 1200, 3), 1.5, False), g['q_map_noflip'])
    assert np.array_equal(T.pointobb_mapping_back(q, (900, 1200, 3), 1.5, True), g['q_mapback'])
    assert np.array_equal(q, g['q_in'])                       # inputs untouched
    hb = g['hq_in']
    assert np.array_equal(np.array([T.hobb2pointobb(h.tolist()) for h in hb]), g['hq_points'])
    assert np.array_equal(T.hobb_mapping(hb, (900, 1200, 3), 1.5, False), g['hq_map_noflip'])
    assert np.array_equal(T.hobb_mapping_back(hb, (900, 1200, 3), 1.5, False), g['hq_mapback_noflip'])
    # (flip=True goes through the minimal rectangle: test_min_area_rect_codecs_without_opencv)
    flipped = T.hobb_mapping(hb, (900, 1200, 3), 1.5, True)
    assert flipped.shape == hb.shape and np.isfinite(flipped).all()


def test_min_area_rect_codecs_without_opencv():
    """pointobb2thetaobb (cv2.minAreaRect in the reference, transforms.py:30-43), thetaobb2hobb (:73-91) and
    hobb_flip (:243-278) restated without cv2: the angle / size convention (theta in [-pi/2, 0), `w` along
    (cos, sin)), the axis-aligned quirk, and the round trip through the corner points."""
    rng = np.random.default_rng(0)
    n = 4000
    t = np.stack([rng.uniform(0, 1000, n), rng.uniform(0, 1000, n), np.exp(rng.uniform(np.log(8), np.log(128), n)),
                  np.exp(rng.uniform(np.log(8), np.log(128), n)), rng.uniform(-np.pi / 2, 0, n)], 1)
    t = t[np.abs(t[:, 2] - t[:, 3]) > 1e-3]                 # (a square has two minimal rectangles)
    r = T.min_area_rect(T.box_points(t))
    assert ((r[:, 4] >= -90.0) & (r[:, 4] < 0.0)).all()
    r[:, 4] *= np.pi / 180.0
    assert np.abs(r - t).max() < 1e-4                       # thetaobb -> pointobb -> thetaobb is the identity
    # any cyclic rotation / reversal of the point list describes the same box
    q = T.box_points(t[:50])
    for perm in ([2, 3, 4, 5, 6, 7, 0, 1], [6, 7, 4, 5, 2, 3, 0, 1]):
        r2 = T.min_area_rect(q[:, perm]); r2[:, 4] *= np.pi / 180.0
        assert np.abs(r2 - t[:50]).max() < 1e-4
    # the box_points twin equals the device op's formula on theta = 0: corners of the upright box
    assert np.allclose(T.box_points([10., 20., 4., 2., 0.]), [8, 21, 8, 19, 12, 19, 12, 21])
    # axis-aligned rectangle 10 wide, 4 high: OpenCV (< 4.5.1) reports angle -90 and the sizes swapped
    assert T.pointobb2thetaobb([0, 0, 10, 0, 10, 4, 0, 4]) == [5.0, 2.0, 4.0, 10.0, -np.pi / 2]
    # a 45-degree rectangle with integer corners: sides 3*sqrt(2) and 4*sqrt(2)
    x, y, w, h, th = T.pointobb2thetaobb([3, 0, 7, 4, 4, 7, 0, 3])
    assert np.allclose([x, y, w, h, th], [3.5, 3.5, 3 * 2 ** 0.5, 4 * 2 ** 0.5, -np.pi / 4])
    # the reference truncates the points to integers first (np.int0)
    assert T.pointobb2thetaobb([0.9, 0.9, 10.9, 0.2, 10.5, 4.7, 0.1, 4.9]) == T.pointobb2thetaobb([0, 0, 10, 0, 10, 4, 0, 4])
    assert T.pointobb2thetaobb([0.5, 0, 10.5, 0, 10.5, 4, 0.5, 4], truncate=False)[0] == 5.5
    # degenerate inputs stay finite
    assert np.isfinite(T.min_area_rect(np.zeros((1, 8)))).all()
    assert np.isfinite(T.min_area_rect(np.array([[0, 0, 5, 5, 5, 5, 0, 0.]]))).all()
    # thetaobb2hobb: first two corners in the sort function's order + the length of the side 4-1
    hb = T.thetaobb2hobb([50., 50., 40., 20., -0.3], T.pointobb_best_point_sort)
    p = T.pointobb_best_point_sort(T.thetaobb2pointobb_host([50., 50., 40., 20., -0.3]))
    assert hb[:4] == p[:4] and abs(hb[4] - 20.0) < 1e-9
    # hobb_flip mirrors about x = (W - 1) / 2: flipping twice returns the box up to the integer truncations
    h0 = np.array([[100., 40., 160., 60., 30.], [20., 20., 80., 20., 25.]])
    once = T.hobb_flip(h0, (300, 400, 3))
    assert once.shape == (2, 5) and np.allclose(once[1], [319., 20., 379., 20., 25.], atol=1.5)
    twice = T.hobb_flip(once, (300, 400, 3))
    assert np.abs(twice - h0).max() < 3.0


def _raster(cx, cy, w, h, th, H=220, W=260):
    ys, xs = np.mgrid[0:H, 0:W]
    dx, dy = xs - cx, ys - cy
    u, v = dx * np.cos(th) + dy * np.sin(th), -dx * np.sin(th) + dy * np.cos(th)
    return ((np.abs(u) <= w / 2) & (np.abs(v) <= h / 2)).astype(np.uint8)


def _rle(mask, compressed):
    """the encoder side of COCO's RLE, for the test only: column-major runs from 0; the compressed form of maskUtils"""
    flat = mask.T.reshape(-1)
    runs, val, c = [], 0, 0
    for b in flat:
        if b != val:
            runs.append(c); c = 0; val = b
        c += 1
    runs.append(c)
    if not compressed:
        return {'size': list(mask.shape), 'counts': runs}
    out = bytearray()
    for i, x in enumerate(runs):
        if i > 2:
            x -= runs[i - 2]
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = not ((x == -1 and (c & 0x10)) or (x == 0 and not (c & 0x10)))
            out.append((c | 0x20 if more else c) + 48)
    return {'size': list(mask.shape), 'counts': bytes(out)}


def test_maskobb2thetaobb_without_opencv_and_pycocotools():
    """maskobb2thetaobb (transforms.py:166-189: maskUtils.decode -> cv2.findContours -> largest contourArea ->
    cv2.minAreaRect) restated on numpy / scipy.ndimage: rasterised rotated rectangles come back to within a pixel in the
    minAreaRect convention, the largest blob wins, holes do not matter, both RLE forms decode, an empty mask gives zeros."""
    for cx, cy, w, h, deg in ((120, 90, 80, 30, -30.0), (100, 110, 40, 90, -62.0), (130, 100, 60, 24, -8.0)):
        m = _raster(cx, cy, w, h, np.radians(deg))
        m[5:9, 5:12] = 1                                            # a second, smaller blob
        m[int(cy) - 1:int(cy) + 2, int(cx) - 1:int(cx) + 2] = 0     # a hole in the large one
        for form in (m, _rle(m, False), _rle(m, True)):
            x, y, ww, hh, th = T.maskobb2thetaobb(form)
            assert -np.pi / 2 <= th < 0.0
            assert abs(x - cx) < 1.0 and abs(y - cy) < 1.0 and abs(ww - w) < 1.5 and abs(hh - h) < 1.5
            assert abs(np.degrees(th) - deg) < 1.5
    # both RLE forms decode to the mask they came from (odd sizes, runs longer than 31 and 1023: multi-character counts)
    rng = np.random.default_rng(1)
    m = (rng.random((37, 53)) < 0.3).astype(np.uint8)
    m[:, 20:45] = 0
    for compressed in (False, True):
        assert np.array_equal(T.rle_decode(_rle(m, compressed)), m)
    big = np.zeros((300, 70), np.uint8)
    big[10:290, 3:60] = 1
    assert np.array_equal(T.rle_decode(_rle(big, True)), big)
    # axis-aligned blob: the minAreaRect quirk (theta = -90 degrees, sizes swapped); pixel centres span 9 x 3
    ax = np.zeros((20, 30), np.uint8)
    ax[5:9, 10:20] = 1
    assert T.maskobb2thetaobb(ax) == [14.5, 6.5, 3.0, 9.0, -np.pi / 2]
    assert T.maskobb2thetaobb(np.zeros((10, 10), np.uint8)) == [0, 0, 0, 0, 0]
    one = np.zeros((10, 10), np.uint8)
    one[4, 7] = 1
    assert T.maskobb2thetaobb(one) == [7.0, 4.0, 0.0, 0.0, -np.pi / 2]
    # the point-set form agrees with the four-point form on the corners of a rectangle
    t = np.array([[50., 40., 30., 12., -0.4]])
    a = T.min_area_rect(T.box_points(t))[0]
    b = T.min_area_rect_points(T.box_points(t).reshape(-1, 2))
    assert np.allclose(a, b, atol=1e-9)


@pytest.mark.parametrize('enc,dim', [('thetaobb', 5), ('pointobb', 8), ('hobb', 5)])
def test_rbbox_target_reproduces_reference(golden, enc, dim):
    """core/rbbox/rbbox_target.py on three images (one without positives, one without negatives)."""
    from aidet_amd.rbbox_target import rbbox_target
    g = golden('obb_codecs.npz')
    ld = lambda k: torch.from_numpy(g[k])
    pos, neg = [ld('t_pos_%d' % k) for k in range(3)], [ld('t_neg_%d' % k) for k in range(3)]
    inds, glab = [ld('t_inds_%d' % k) for k in range(3)], [ld('t_labels_%d' % k) for k in range(3)]
    gts = [ld('t_gt_%s_%d' % (enc, k)) for k in range(3)]
    means, stds = {'thetaobb': ([0.] * 5, [.1, .1, .2, .2, .1]), 'pointobb': (g['p_means'].tolist(), g['p_stds'].tolist()),
                   'hobb': (g['h_means'].tolist(), g['h_stds'].tolist())}[enc]
    res = rbbox_target(pos, neg, inds, gts, glab, dict(encode=enc), means, stds, out_dim_reg=dim)
    for name, got in zip(('labels', 'label_weights', 'targets', 'weights'), res):
        want = g['t_%s_%s' % (enc, name)]
        assert got.numpy().dtype == want.dtype and np.array_equal(got.numpy(), want), name
    per = rbbox_target(pos, neg, inds, gts, glab, dict(encode=enc), means, stds, out_dim_reg=dim, concat=False)
    assert [len(x) for x in per[0]] == g['t_%s_split' % enc].tolist()
