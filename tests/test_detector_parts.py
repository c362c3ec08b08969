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


@pytest.mark.gpu
def test_detector_tile_ops_in_situ_against_the_oracle():
    """BASELINE configs[3], op by op ON THE TENSORS OF THE REAL TILE: the 1024 x 1024 tile is run
    through the conv trunk once, then every operator stage is fed the real intermediate tensors and compared with the
    oracle / the tensor-op specification:
      RPN      selection kernel vs torch.topk on the 196 k + 49 k + 12 k + 3 k + 768 anchors' logits; grouped 5-level
               nms(0.7) on the decoded boxes: index-exact vs oracle.hbb_nms(groups=level)
      RoIAlign V1 on the 2000 real proposals, level by level: bit-equal vs oracle.roi_align_forward
      bbox     candidates of the fused kernel = the tensor-op candidates (softmax, > 0.05, delta2bbox); class-wise
               nms(0.5): index-exact vs oracle.hbb_nms(groups=class); detections of fused and tensor-op path agree
      rbbox    keep re-use rows = the tensor-op path's; true rotated NMS on the decoded theta-OBBs: index-exact vs
               oracle.rnms(groups=class)
    """
    import ctypes

    import oracle
    from aidet_amd import _lib, nms_wrapper, post_processing as pp
    from aidet_amd.bbox_transforms import bbox2roi, delta2bbox
    from aidet_amd.models import build_detector, demo_inputs
    from aidet_amd.rbbox_transforms import delta2thetaobb
    from aidet_amd.rpn import rpn_proposals
    dev = torch.device('cuda:0')
    model = build_detector(dev)
    img, meta = demo_inputs(dev)
    cfg = model.test_cfg
    with torch.no_grad():
        x = model.extract_feat(img)
        cls_scores, bbox_preds = model.rpn_head(x)
        cls = [c[0].contiguous() for c in cls_scores]
        reg = [r[0].contiguous() for r in bbox_preds]
        anchors = model.rpn_head.anchors([tuple(c.shape[-2:]) for c in cls], dev)
        # ---- RPN selection + decode
        L = _lib.lib()
        n = len(cls)
        per_level = [min(2000, c.numel()) for c in cls]
        N = sum(per_level)
        dets = torch.empty((N, 5), device=dev)
        levels = torch.empty(N, dtype=torch.int32, device=dev)
        rows = ctypes.c_int64(0)
        _lib.check(L.obb_rpn_select_decode_f32((ctypes.c_void_p * n)(*[c.data_ptr() for c in cls]), (ctypes.c_void_p * n)(*[r.data_ptr() for r in reg]),
                                               (ctypes.c_void_p * n)(*[a.data_ptr() for a in anchors]), (ctypes.c_int * n)(*[c.size(1) for c in cls]),
                                               (ctypes.c_int * n)(*[c.size(2) for c in cls]), n, 3, 1, 2000, (ctypes.c_float * 4)(0, 0, 0, 0),
                                               (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 1024, 1024, dets.data_ptr(), levels.data_ptr(),
                                               ctypes.byref(rows), _lib.current_stream_handle(dev)))
        assert rows.value == N == 2000 * 4 + 768
        start = 0
        for l, (c, r, a) in enumerate(zip(cls, reg, anchors)):
            flat = c.permute(1, 2, 0).reshape(-1)
            m = per_level[l]
            got = dets[start:start + m]
            if flat.numel() > m:
                top_s, top_i = flat.sigmoid().topk(m)
                assert torch.allclose(got[:, 4], top_s, atol=1e-6)                      # the same scores, descending
                want = delta2bbox(a[top_i], r.permute(1, 2, 0).reshape(-1, 4)[top_i], (0, 0, 0, 0), (1, 1, 1, 1), (1024, 1024, 3))
                if bool((top_s[:-1] > top_s[1:]).all()):     # (equal scores: torch.topk's order among them is unspecified)
                    assert float((got[:, :4] - want).abs().max()) <= 2e-4
            else:
                want = delta2bbox(a, r.permute(1, 2, 0).reshape(-1, 4), (0, 0, 0, 0), (1, 1, 1, 1), (1024, 1024, 3))
                assert float((got[:, :4] - want).abs().max()) <= 2e-4 and torch.allclose(got[:, 4], flat.sigmoid(), atol=1e-6)
            assert bool((levels[start:start + m] == l).all())
            start += m
        # ---- grouped 5-level NMS on the real decoded boxes: index-exact
        keep, _ = nms_wrapper._run_nms(dets, _lib.KIND_HBB, 0.7, levels)
        ref = oracle.hbb_nms(dets.cpu().numpy(), 0.7, groups=levels.cpu().numpy())
        assert np.array_equal(keep.cpu().numpy(), ref) and 2000 < len(ref) < N
        # ---- proposals: fused == tensor-op path
        prop = rpn_proposals(cls, reg, anchors, (1024, 1024, 3), cfg['rpn'], fused=True)
        prop_ref = rpn_proposals(cls, reg, anchors, (1024, 1024, 3), cfg['rpn'], fused=False)
        from test_round3_gpu import same_rows_up_to_score_ties
        assert prop.shape == prop_ref.shape == (2000, 5) and same_rows_up_to_score_ties(prop, prop_ref)
        rois = bbox2roi([prop])
        # ---- RoIAlign V1 on the real RoIs, level by level, bit-equal with the oracle
        ext = model.bbox_roi_extractor
        feats = ext(x[:4], rois)
        lv = ext.map_roi_levels(rois, 4).cpu().numpy()
        rn = rois.cpu().numpy()
        fn = feats.cpu().numpy()
        assert len(set(lv.tolist())) >= 3
        for l in range(4):
            idx = np.nonzero(lv == l)[0][:160]          # (160 RoIs per level: a few seconds of oracle time)
            if len(idx) == 0:
                continue
            want = oracle.roi_align_forward(x[l].cpu().numpy(), rn[idx], 7, 1.0 / (4 << l), 2, version=1)
            assert np.array_equal(fn[idx], want), 'level %d' % l
        # ---- bbox head on the real features
        cls_score, bbox_pred = model.bbox_head(feats)
        cd, crows, clab, counts = pp.select_decode_candidates(rois, cls_score, bbox_pred, 4, 0.05, (0, 0, 0, 0), model.bbox_target_stds, (1024, 1024, 3))
        scores = torch.softmax(cls_score, 1)
        boxes = delta2bbox(rois[:, 1:], bbox_pred, (0., 0., 0., 0.), model.bbox_target_stds, (1024, 1024, 3))
        mask = (scores[:, 1:] > 0.05).t()
        wl, wr = mask.nonzero(as_tuple=True)
        edge = ((scores[:, 1:] - 0.05).abs() < 1e-6).any()
        if not bool(edge):                              # (no probability within 1e-6 of the threshold: identical lists)
            assert torch.equal(crows.long(), wr) and torch.equal(clab.long(), wl) and counts == mask.sum(1).tolist()
            want = boxes.view(-1, 16, 4)[wr, wl + 1]
            assert float((cd[:, :4] - want).abs().max()) <= 2e-4 and torch.allclose(cd[:, 4], scores[wr, wl + 1], atol=1e-6)
        keep, _ = nms_wrapper._run_nms(cd, _lib.KIND_HBB, 0.5, clab)
        ref = oracle.hbb_nms(cd.cpu().numpy(), 0.5, groups=clab.cpu().numpy())
        assert np.array_equal(keep.cpu().numpy(), ref) and len(ref) > 100
        db, dl, sel = pp.detect_horizontal_fused(rois, cls_score, bbox_pred, (1024, 1024, 3), 1.0, False, cfg['rcnn'], (0, 0, 0, 0), model.bbox_target_stds)
        rb, rl, cls_inds, keep_inds = pp.multiclass_nms_with_index(boxes, scores, 0.05, cfg['rcnn']['nms'], 1000)

        def same_detections(got, got_l, want, want_l, pool, pool_l):
            # a random-init head saturates: hundreds of heavily overlapping detections share one score.  WHICH of them the
            # cut at max_per_img keeps (and in what order) is the sort's business, and boxes decoded by the kernel differ
            # from torch's by ~1e-4 px, enough to flip an IoU that sits on the threshold.  The NMS decisions themselves are
            # checked index-exact against the oracle above; here: the same score sequence, and every returned (row, label)
            # is one of the NMS survivors
            assert got.shape == want.shape and torch.allclose(got[:, -1], want[:, -1], atol=1e-6)
            assert bool((got[:-1, -1] >= got[1:, -1]).all()) or got.size(0) <= 1
            have = {tuple(np.round(r, 2)) + (int(l),) for r, l in zip(pool.cpu().numpy(), pool_l.cpu().numpy())}
            assert all(tuple(np.round(r, 2)) + (int(l),) in have for r, l in zip(got.cpu().numpy(), got_l.cpu().numpy()))
        same_detections(db, dl, rb, rl, cd[keep], clab[keep])
        # ---- rotated head: keep re-use and the true rotated NMS
        rfeats = model.rbbox_roi_extractor(x[:4], rois)
        rcls, rpred = model.rbbox_head(rfeats)
        fr, fl = pp.detect_rotated_fused(rois, rcls, rpred, (1024, 1024, 3), 1.0, False, cfg['rbbox'], (0,) * 5, model.rbbox_target_stds, selection=sel)
        wr_, wl_ = pp.get_det_rbboxes(rois, rcls, rpred, (1024, 1024, 3), 1.0, cfg=cfg['rbbox'], target_stds=model.rbbox_target_stds,
                                      bbox_cls_inds=cls_inds, bbox_keep_inds=keep_inds, rotated_nms=False)
        pool = pp.gather_decode(rois, rcls, rpred, 5, sel[0], sel[1], sel[2], (0,) * 5, model.rbbox_target_stds)
        same_detections(fr, fl, wr_, wl_, pool, sel[1][sel[2]])
        td, trows, tlab, _ = pp.select_decode_candidates(rois, rcls, rpred, 5, 0.05, (0,) * 5, model.rbbox_target_stds)
        tb = delta2thetaobb(rois[:, 1:], rpred, (0,) * 5, model.rbbox_target_stds).view(-1, 16, 5)[trows.long(), tlab.long() + 1]
        assert float((td[:, :5] - tb).abs().max()) <= 2e-4
        rkeep, _ = nms_wrapper._run_nms(td, _lib.KIND_RBOX, 0.5, tlab)
        ref = oracle.rnms(td.cpu().numpy(), 0.5, groups=tlab.cpu().numpy())
        assert np.array_equal(rkeep.cpu().numpy(), ref) and len(ref) > 100
        fr2, fl2 = pp.detect_rotated_fused(rois, rcls, rpred, (1024, 1024, 3), 1.0, False, cfg['rbbox'], (0,) * 5, model.rbbox_target_stds)
        wr2, wl2 = pp.get_det_rbboxes(rois, rcls, rpred, (1024, 1024, 3), 1.0, cfg=cfg['rbbox'], target_stds=model.rbbox_target_stds, rotated_nms=True)
        same_detections(fr2, fl2, wr2, wl2, td[rkeep], tlab[rkeep])


@pytest.mark.gpu
@pytest.mark.parametrize('rotated', [False, True])
def test_detector_fused_stages_equal_the_tensor_op_stages(rotated):
    """The whole tile through the fused selection kernels and through the tensor-op restatement: the same detections."""
    from aidet_amd.models import build_detector, demo_inputs
    dev = torch.device('cuda:0')
    img, meta = demo_inputs(dev)
    a = build_detector(dev, rotated_nms=rotated, fused=True).simple_test(img, meta)
    b = build_detector(dev, rotated_nms=rotated, fused=False).simple_test(img, meta)
    # (per class: the same number of detections with the same scores; rows of equal score may be other members of the tie)
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert x.shape == y.shape
        if x.size:
            assert np.allclose(np.sort(x[:, -1]), np.sort(y[:, -1]), atol=1e-6)
