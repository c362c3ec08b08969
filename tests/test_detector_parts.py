"""Config 4 (Faster-RCNN-OBB R50-FPN), everything between the convolutions and the results,
against values captured from the reference's own classes (tests/golden/detector_parts.npz,
tests/golden/make_golden.py: AnchorGenerator, delta2bbox, bbox2roi, RPNHead.get_bboxes on the
reference's CPU nms, SingleRoIExtractor.map_roi_levels / roi_rescale, BBoxHead.get_det_bboxes).
CPU tests inject the oracle's HBB NMS for the op; the GPU test runs the HIP op and the model."""
import numpy as np
import pytest
import torch

import oracle


def _oracle_nms(dets, iou_thr, device_id=None, groups=None):
    keep = torch.from_numpy(oracle.hbb_nms(dets.detach().cpu().numpy(), float(iou_thr),
                                           groups=None if groups is None else groups.cpu().numpy()))
    return dets[keep], keep


def test_anchor_generator_reproduces_reference(golden):
    from aidet_amd.anchors import AnchorGenerator
    g = golden('detector_parts.npz')
    strides = g['anchor_strides'].tolist()
    for i, s in enumerate(strides):
        gen = AnchorGenerator(s, [8], [0.5, 1.0, 2.0])
        assert gen.num_base_anchors == 3
        assert np.array_equal(gen.base_anchors.numpy(), g['base_anchors_%d' % i])
        assert np.array_equal(gen.grid_anchors((3, 5), s, device='cpu').numpy(), g['grid_small_%d' % i])
        full = gen.grid_anchors((1024 // s, 1024 // s), s, device='cpu').numpy()   # the 1024^2 tile of config 4
        assert len(full) == int(g['grid_tile_len_%d' % i])
        assert np.array_equal(full[g['grid_tile_rows_%d' % i]], g['grid_tile_sample_%d' % i])
        assert np.array_equal(full.astype(np.float64).sum(0), g['grid_tile_sum_%d' % i])
    assert np.array_equal(AnchorGenerator(9, [1.], [1.]).grid_anchors((2, 2), device='cpu').numpy(), g['doc_grid'])
    assert g['doc_grid'].tolist() == [[0, 0, 8, 8], [16, 0, 24, 8], [0, 16, 8, 24], [16, 16, 24, 24]]   # anchor_generator.py:10-14
    g2 = AnchorGenerator(16, [2., 4.], [0.5, 2.0], scale_major=False, ctr=(3.0, 5.0))
    assert np.array_equal(g2.base_anchors.numpy(), g['ctr_base_anchors'])


def test_delta2bbox_and_bbox2roi_reproduce_reference(golden):
    from aidet_amd.bbox_transforms import bbox2result, bbox2roi, delta2bbox
    g = golden('detector_parts.npz')
    got = delta2bbox(torch.from_numpy(g['d2b_doc_rois']), torch.from_numpy(g['d2b_doc_deltas']), max_shape=(32, 32))
    assert np.array_equal(got.numpy(), g['d2b_doc_out'])
    assert np.allclose(got.numpy(), [[0, 0, 1, 1], [0.2817, 0.2817, 4.7183, 4.7183], [0, 0.6321, 7.3891, 0.3679],
                                     [5.8967, 2.9251, 5.5033, 3.2749]], atol=5e-5)     # transforms.py:72-76
    R, D = torch.from_numpy(g['d2b_rois']), torch.from_numpy(g['d2b_deltas'])
    assert np.array_equal(delta2bbox(R, D, [0., 0., 0., 0.], [0.1, 0.1, 0.2, 0.2], (1024, 1024, 3)).numpy(), g['d2b_out'])
    assert np.array_equal(delta2bbox(R, D[:, :4], [0.1, -0.1, 0.05, 0.], [1., 1., 1., 1.]).numpy(), g['d2b_out_noclip'])
    rois = bbox2roi([R[:3, :], R[:0, :], torch.cat([R[3:5], torch.ones(2, 1)], 1)])
    assert np.array_equal(rois.numpy(), g['bbox2roi'])
    res = bbox2result(torch.cat([R[:6], torch.ones(6, 1)], 1), torch.tensor([0, 2, 2, 14, 0, 3]), 16)
    assert len(res) == 15 and [len(r) for r in res][:4] == [2, 0, 2, 1] and res[0].shape == (2, 5)
    assert [r.shape for r in bbox2result(torch.zeros(0, 5), torch.zeros(0), 16)] == [(0, 5)] * 15


def _rpn_inputs(g):
    cls = [torch.from_numpy(g['rpn_cls_%d' % i])[0] for i in range(5)]
    reg = [torch.from_numpy(g['rpn_reg_%d' % i])[0] for i in range(5)]
    return cls, reg


def _rpn_cfg(row):
    return dict(nms_across_levels=bool(row[0]), nms_pre=int(row[1]), nms_post=int(row[2]), max_num=int(row[3]),
                nms_thr=float(row[4]), min_bbox_size=float(row[5]))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_rpn_proposals_reproduce_reference(golden, tag):
    """RPNHead.get_bboxes_single (rpn_head.py:55-104): the five per-level NMS calls of the
    reference as ONE grouped call -- identical proposals, identical order."""
    from aidet_amd.anchors import AnchorGenerator
    from aidet_amd.rpn import rpn_proposals
    g = golden('detector_parts.npz')
    cls, reg = _rpn_inputs(g)
    strides = g['anchor_strides'].tolist()
    anchors = [AnchorGenerator(s, [8], [0.5, 1.0, 2.0]).grid_anchors(c.shape[-2:], s, device='cpu')
               for s, c in zip(strides, cls)]
    got = rpn_proposals(cls, reg, anchors, (256, 256, 3), _rpn_cfg(g['rpn_%s_cfg' % tag]), nms_op=_oracle_nms)
    want = g['rpn_%s_proposals' % tag]
    assert got.shape == want.shape and np.array_equal(got.numpy(), want)


def test_roi_extractor_level_mapping_reproduces_reference(golden):
    from aidet_amd.roi_layers import SingleRoIExtractor
    g = golden('detector_parts.npz')
    ext = SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=2), 256, [4, 8, 16, 32])
    assert len(ext.roi_layers) == 4 and ext.num_inputs == 4 and ext.roi_layers[2].spatial_scale == 1 / 16
    rois = torch.from_numpy(g['lvl_rois'])
    lv = ext.map_roi_levels(rois, 4)
    assert np.array_equal(lv.numpy(), g['lvl_target'])
    assert lv[:8].tolist() == [1, 0, 2, 1, 3, 2, 0, 3]       # scale 112 / 224 / 448 = the level borders
    assert np.array_equal(ext.roi_rescale(rois, 1.25).numpy(), g['lvl_rescaled'])


def test_bbox_head_decode_reproduces_reference(golden):
    """BBoxHead.get_det_bboxes(cfg=None) (bbox_head.py:139-176) = softmax + delta2bbox + rescale."""
    from aidet_amd.bbox_transforms import delta2bbox
    g = golden('detector_parts.npz')
    R, D = torch.from_numpy(g['d2b_rois']), torch.from_numpy(g['d2b_deltas'])
    scores = torch.softmax(torch.from_numpy(g['head_cls_score']), dim=1)
    bboxes = delta2bbox(R, D, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), (1024, 1024, 3)) / 2.0
    assert np.array_equal(scores.numpy(), g['head_scores']) and np.array_equal(bboxes.numpy(), g['head_bboxes'])


def test_model_builds_with_the_configured_shapes():
    """config dota_v002_theta_obb_r50_v1_train.py:11-76: channels and head widths."""
    from aidet_amd.models import TEST_CFG, build_detector, demo_inputs
    m = build_detector('cpu')
    assert TEST_CFG['rpn']['nms_thr'] == 0.7 and TEST_CFG['rcnn']['max_per_img'] == 1000
    assert m.bbox_head.fc_reg.out_features == 64 and m.rbbox_head.fc_reg.out_features == 80
    assert m.bbox_head.fc_cls.out_features == 16 and m.bbox_head.shared_fcs[0].in_features == 256 * 49
    img, meta = demo_inputs('cpu', 128)
    with torch.no_grad():
        feats = m.extract_feat(img)
        cls, reg = m.rpn_head(feats)
    assert [tuple(f.shape[1:]) for f in feats] == [(256, 32, 32), (256, 16, 16), (256, 8, 8), (256, 4, 4), (256, 2, 2)]
    assert [c.shape[1] for c in cls] == [3] * 5 and [r.shape[1] for r in reg] == [12] * 5
    a = m.rpn_head.anchors([tuple(c.shape[-2:]) for c in cls], 'cpu')
    assert [len(x) for x in a] == [32 * 32 * 3, 16 * 16 * 3, 8 * 8 * 3, 4 * 4 * 3, 2 * 2 * 3]


def _aug_metas(g):
    return [dict(img_shape=(int(r[0]), int(r[1]), 3), scale_factor=float(r[2]), flip=bool(r[3])) for r in g['aug_metas']]


def test_bbox_flip_mapping_and_aug_merges_reproduce_reference(golden):
    """core/bbox/transforms.py:114-146 and merge_augs.py:46-78 (no NMS in these)."""
    from aidet_amd.bbox_transforms import bbox_flip, bbox_mapping, bbox_mapping_back
    from aidet_amd.merge_augs import merge_aug_bboxes, merge_aug_proposals, merge_aug_scores
    g = golden('detector_parts.npz')
    fb = torch.from_numpy(g['aug_flip_in'])
    assert np.array_equal(bbox_flip(fb, (300, 400, 3)).numpy(), g['aug_flip_out'])
    assert np.array_equal(bbox_flip(g['aug_flip_in'], (300, 400, 3)), g['aug_flip_out'])      # ndarray in, ndarray out
    assert np.array_equal(bbox_mapping(fb, (450, 600, 3), 1.5, True).numpy(), g['aug_map_out'])
    assert np.array_equal(bbox_mapping(fb, (450, 600, 3), 1.5, False).numpy(), g['aug_map_noflip'])
    assert np.array_equal(bbox_mapping_back(fb, (450, 600, 3), 1.5, True).numpy(), g['aug_mapback_out'])
    assert np.array_equal(fb.numpy(), g['aug_flip_in'])                                        # inputs untouched
    metas = _aug_metas(g)
    ab = [torch.from_numpy(g['aug_bboxes_%d' % k]) for k in range(3)]
    asc = [torch.from_numpy(g['aug_scores_%d' % k]) for k in range(3)]
    mb, ms = merge_aug_bboxes(ab, asc, [[m] for m in metas], None)
    assert np.array_equal(mb.numpy(), g['aug_bboxes_merged']) and np.array_equal(ms.numpy(), g['aug_scores_merged'])
    assert np.array_equal(merge_aug_bboxes(ab, None, [[m] for m in metas]).numpy(), g['aug_bboxes_only'])
    assert np.array_equal(merge_aug_scores(asc).numpy(), g['aug_scores_mean_t'])
    assert np.array_equal(merge_aug_scores([a.numpy() for a in asc]), g['aug_scores_mean_np'])
    # merge_aug_proposals with the oracle's op injected (the HIP op: GPU test below)
    props = [torch.from_numpy(g['aug_props_%d' % k]) for k in range(3)]
    for cfg, key in ((dict(nms_thr=0.7, max_num=120), 'aug_props_merged'), (dict(nms_thr=0.5, max_num=100000), 'aug_props_merged_all')):
        got = merge_aug_proposals(props, metas, cfg, nms_op=_oracle_nms).numpy()
        assert np.array_equal(got, g[key]), key


# ---- on the GPU: the HIP ops inside the same functions, then the whole detector -----------------
@pytest.mark.gpu
def test_merge_aug_proposals_on_gpu_reproduces_reference(golden):
    """aug_test_rpn's merge (test_mixins.py:37-59 -> merge_augs.py:8-43) on the HIP nms."""
    from aidet_amd.merge_augs import merge_aug_proposals
    g = golden('detector_parts.npz')
    props = [torch.from_numpy(g['aug_props_%d' % k]).cuda() for k in range(3)]
    for cfg, key in ((dict(nms_thr=0.7, max_num=120), 'aug_props_merged'), (dict(nms_thr=0.5, max_num=100000), 'aug_props_merged_all')):
        got = merge_aug_proposals(props, _aug_metas(g), cfg).cpu().numpy()
        # same survivors in the same order; torch divides by a scalar on the GPU as a multiplication by
        # its reciprocal, so mapped-back coordinates may differ from the CPU reference in the last ulp
        assert got.shape == g[key].shape and np.array_equal(got[:, 4], g[key][:, 4]), key
        assert np.abs(got - g[key]).max() <= 1e-4, key


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_rpn_proposals_on_gpu_reproduce_reference(golden, tag):
    from aidet_amd.anchors import AnchorGenerator
    from aidet_amd.rpn import rpn_proposals
    g = golden('detector_parts.npz')
    cls, reg = _rpn_inputs(g)
    dev = torch.device('cuda:0')
    strides = g['anchor_strides'].tolist()
    anchors = [AnchorGenerator(s, [8], [0.5, 1.0, 2.0]).grid_anchors(c.shape[-2:], s, device=dev)
               for s, c in zip(strides, cls)]
    got = rpn_proposals([c.to(dev) for c in cls], [r.to(dev) for r in reg], anchors, (256, 256, 3),
                        _rpn_cfg(g['rpn_%s_cfg' % tag])).cpu().numpy()
    want = g['rpn_%s_proposals' % tag]
    assert got.shape == want.shape
    # the keep decisions are the reference's (same boxes in the same order); the decoded
    # coordinates may differ in the last ulp between the CPU and the GPU's exp / sigmoid
    assert np.abs(got - want).max() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize('rotated', [False, True])
def test_detector_tile_forward(rotated):
    """BASELINE configs[3]: one 1024 x 1024 tile through the whole detector, random init (seed 0).
    Checks the contract of RBBoxRCNN.simple_test (rbbox_cnn.py:214-253): 15 per-class arrays
    (n, 5) and (n, 6), at most max_per_img detections, determinism, and the relation between
    the two post-processing modes."""
    from aidet_amd.models import build_detector, demo_inputs
    dev = torch.device('cuda:0')
    model = build_detector(dev, rotated_nms=rotated)
    img, meta = demo_inputs(dev)
    bb, rb = model.simple_test(img, meta)
    assert len(bb) == len(rb) == 15
    assert all(b.ndim == 2 and b.shape[1] == 5 and b.dtype == np.float32 for b in bb)
    assert all(r.ndim == 2 and r.shape[1] == 6 for r in rb)
    nb, nr = sum(len(b) for b in bb), sum(len(r) for r in rb)
    assert 0 < nb <= 1000 and 0 < nr <= 1000
    for b in bb:
        if len(b):
            assert b[:, :4].min() >= 0 and b[:, :4].max() <= 1023 and b[:, 4].min() > 0.05
    bb2, rb2 = model.simple_test(img, meta)
    assert all(np.array_equal(x, y) for x, y in zip(bb, bb2)) and all(np.array_equal(x, y) for x, y in zip(rb, rb2))
    if not rotated:
        # the reference's keep re-use (rbbox_head.py:287-293): as many rotated boxes as horizontal
        # ones (scored by the rotated head's own classifier)
        assert nb == nr
    # stage timers of the bench hook
    model.timers = {}
    model.simple_test(img, meta)
    assert {'backbone_fpn', 'rpn_decode_nms', 'roi_align_bbox', 'bbox_decode_nms', 'roi_align_rbbox',
            'rbbox_decode_nms'} <= set(model.timers)
