"""CPU tests: theta-OBB codec vs values captured from the reference
(tests/golden/thetaobb_codec.npz, anchor A5) and host-side RoI-layer mappings."""
import numpy as np
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
