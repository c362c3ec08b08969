"""Generate tests/golden/*.npz by RUNNING the reference (jwwangchn/aidet).

Runs only in the build container: needs /root/reference and oracle/_ref/nms_cpu.so
(python oracle/build_ref.py).  The fixtures are DATA (inputs + the reference's
outputs); no reference source is stored.  Re-run:  python tests/golden/make_golden.py

What is captured (SURVEY.md 8c anchors):
  hbb_iou.npz        A1  mmdet/core/bbox/geometry.py:4-88   bbox_overlaps
  hbb_nms.npz        A2  mmdet/ops/nms/src/nms_cpu.cpp:5-68 via nms_wrapper.py:7-60
  multiclass.npz     A3  core/post_processing/bbox_nms.py:6-76, rbbox_nms.py:6-119
  multiclass_soft.npz    the same two functions with nms_cfg type='soft_nms' (on multiclass.npz's inputs)
  thetaobb_codec.npz A5  core/rbbox/transforms.py:191-203,280-293,321-395
  assigner.npz           core/bbox/assigners/max_iou_assigner.py:8-202 (MaxIoUAssigner)
  soft_nms.npz           mmdet/ops/nms/src/nms_cpu.cpp:71-210 via nms_wrapper.py:63-118
  hbb_nms_ties.npz       nms_cpu.cpp:5-68 on INTEGER boxes (IoU == thr happens: `>=` decides)
  detector_parts.npz     config 4, everything between the convolutions and the results:
                         core/anchor/anchor_generator.py:4-98, core/bbox/transforms.py:34-168
                         (delta2bbox, bbox2roi), models/anchor_heads/rpn_head.py:55-104 through
                         anchor_head.py:254-283 (RPNHead.get_bboxes on the reference's CPU nms),
                         models/roi_extractors/single_level.py:54-87 (map_roi_levels, roi_rescale),
                         models/bbox_heads/bbox_head.py:139-176 (get_det_bboxes, cfg=None)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    from oracle import build_ref
    build_ref.build()
    nms_cpu = build_ref.load_ref()
    assert nms_cpu is not None
    for pkg in ('mmdet', 'mmdet.ops', 'mmdet.ops.nms'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules['mmdet.ops.nms.nms_cpu'] = nms_cpu
    sys.modules['mmdet.ops.nms'].nms_cpu = nms_cpu
    cuda_stub = types.ModuleType('mmdet.ops.nms.nms_cuda')
    sys.modules['mmdet.ops.nms.nms_cuda'] = cuda_stub
    sys.modules['mmdet.ops.nms'].nms_cuda = cuda_stub
    nms_wrapper = _load('mmdet.ops.nms.nms_wrapper', REF + '/mmdet/ops/nms/nms_wrapper.py')
    sys.modules['mmdet.ops.nms'].nms_wrapper = nms_wrapper
    geometry = _load('ref_geometry', REF + '/mmdet/core/bbox/geometry.py')
    bbox_nms = _load('ref_bbox_nms', REF + '/mmdet/core/post_processing/bbox_nms.py')
    rbbox_nms = _load('ref_rbbox_nms', REF + '/mmdet/core/post_processing/rbbox_nms.py')
    # transforms.py imports cv2 / pycocotools at module level; the functions
    # captured here never call them.
    for stub in ('cv2', 'pycocotools', 'pycocotools.mask', 'mmcv'):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    transforms = _load('ref_rbbox_transforms', REF + '/mmdet/core/rbbox/transforms.py')
    return nms_wrapper, geometry, bbox_nms, rbbox_nms, transforms


def rand_hbb(rng, n, canvas, smin=8, smax=128, integer=False):
    x1 = rng.uniform(0, canvas, n)
    y1 = rng.uniform(0, canvas, n)
    w = np.exp(rng.uniform(np.log(smin), np.log(smax), n))
    h = np.exp(rng.uniform(np.log(smin), np.log(smax), n))
    b = np.stack([x1, y1, x1 + w, y1 + h], 1)
    if integer:
        b = np.round(b)
    return b.astype(np.float32)


def main():
    nms_wrapper, geometry, bbox_nms, rbbox_nms, transforms = load_reference()
    rng = np.random.default_rng(20261003)

    # ---- A1: HBB IoU ---------------------------------------------------
    out = {}
    b1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    b2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    out['doc_b1'] = b1.numpy()
    out['doc_b2'] = b2.numpy()
    out['doc_iou'] = geometry.bbox_overlaps(b1, b2).numpy()
    r1 = torch.from_numpy(rand_hbb(rng, 200, 512))
    r2 = torch.from_numpy(rand_hbb(rng, 300, 512))
    out['rand_b1'] = r1.numpy()
    out['rand_b2'] = r2.numpy()
    out['rand_iou'] = geometry.bbox_overlaps(r1, r2).numpy()
    out['rand_iof'] = geometry.bbox_overlaps(r1, r2, mode='iof').numpy()
    out['rand_iou_aligned'] = geometry.bbox_overlaps(r1, r2[:200], is_aligned=True).numpy()
    out['rand_iof_aligned'] = geometry.bbox_overlaps(r1, r2[:200], mode='iof', is_aligned=True).numpy()
    i1 = torch.from_numpy(rand_hbb(rng, 64, 96, integer=True))
    i2 = torch.from_numpy(rand_hbb(rng, 64, 96, integer=True))
    out['int_b1'] = i1.numpy()
    out['int_b2'] = i2.numpy()
    out['int_iou'] = geometry.bbox_overlaps(i1, i2).numpy()
    empty = torch.FloatTensor([])
    nonempty = torch.FloatTensor([[0, 0, 10, 9]])
    out['empty_shapes'] = np.array([
        tuple(geometry.bbox_overlaps(empty, nonempty).shape),
        tuple(geometry.bbox_overlaps(nonempty, empty).shape),
        tuple(geometry.bbox_overlaps(empty, empty).shape)], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'hbb_iou.npz'), **out)

    # ---- A2: HBB NMS (reference nms_cpu.cpp through nms_wrapper.nms) -----
    out = {}
    t4 = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9],
                   [35.3, 11.5, 39.9, 14.5, 0.4], [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    d7 = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9],
                   [49.2, 31.8, 51.0, 35.4, 0.5], [35.1, 11.5, 39.1, 15.7, 0.5],
                   [35.6, 11.8, 39.3, 14.2, 0.5], [35.3, 11.5, 39.9, 14.5, 0.4],
                   [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    out['test4_dets'] = t4
    out['test4_keep'] = nms_wrapper.nms(t4, 0.7)[1]
    out['doc7_dets'] = d7
    out['doc7_keep'] = nms_wrapper.nms(d7, 0.7)[1]
    # witness of `>=` (nms_cpu.cpp:56): identical boxes at thr 1.0
    two = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8]], dtype=np.float32)
    out['ge_dets'] = two
    out['ge_keep'] = nms_wrapper.nms(two, 1.0)[1]
    for tag, n, canvas, thr in (('a', 1000, 256, 0.5), ('b', 3000, 512, 0.3),
                                ('c', 3000, 512, 0.7), ('d', 5000, 2048, 0.5)):
        boxes = rand_hbb(rng, n, canvas)
        score = ((rng.permutation(n) + 0.5) / n).astype(np.float32)
        dets = np.concatenate([boxes, score[:, None]], 1)
        keep = nms_wrapper.nms(dets, thr)[1]
        out['rand_%s_dets' % tag] = dets
        out['rand_%s_thr' % tag] = np.float32(thr)
        out['rand_%s_keep' % tag] = keep
    np.savez_compressed(os.path.join(HERE, 'hbb_nms.npz'), **out)

    # ---- A3: class-wise orchestration ------------------------------------
    out = {}
    torch.manual_seed(0)
    n, ncls = 200, 4
    hb = torch.from_numpy(rand_hbb(rng, n, 128, 16, 64))
    multi_bboxes = (hb[:, None, :] + torch.randn(n, ncls, 4) * 2).reshape(n, ncls * 4)
    multi_scores = torch.softmax(torch.randn(n, ncls) * 2, dim=1)
    nms_cfg = dict(type='nms', iou_thr=0.5)
    bboxes, labels, cls_inds, keep_inds = rbbox_nms.multiclass_nms_with_index(
        multi_bboxes, multi_scores, 0.05, nms_cfg, 100)
    cx = (hb[:, 0] + hb[:, 2]) * 0.5
    cy = (hb[:, 1] + hb[:, 3]) * 0.5
    w = hb[:, 2] - hb[:, 0]
    h = hb[:, 3] - hb[:, 1]
    th = -torch.rand(n) * np.pi / 2
    multi_rb = torch.stack([cx, cy, w, h, th], 1)[:, None, :].repeat(1, ncls, 1)
    multi_rb = (multi_rb + torch.randn(n, ncls, 5) * torch.tensor([2., 2., 2., 2., .1])).reshape(n, ncls * 5)
    keep_copy = [k.clone() for k in keep_inds]
    rb, rl = rbbox_nms.thetaobb_nms_by_bbox_nms(multi_rb, multi_scores, cls_inds, keep_copy, 100, 5)
    out['multi_bboxes'] = multi_bboxes.numpy()
    out['multi_scores'] = multi_scores.numpy()
    out['multi_rbboxes'] = multi_rb.numpy()
    out['score_thr'] = np.float32(0.05)
    out['iou_thr'] = np.float32(0.5)
    out['max_num'] = np.int64(100)
    out['idx_bboxes'] = bboxes.numpy()
    out['idx_labels'] = labels.numpy()
    out['idx_keep_lens'] = np.array([len(k) for k in keep_inds], dtype=np.int64)
    for i, k in enumerate(keep_inds):
        out['idx_keep_%d' % i] = k.numpy()
    out['rb_bboxes'] = rb.numpy()
    out['rb_labels'] = rl.numpy()
    mb, ml = bbox_nms.multiclass_nms(multi_bboxes, multi_scores, 0.05, nms_cfg, 100)
    out['mc_bboxes'] = mb.numpy()
    out['mc_labels'] = ml.numpy()
    np.savez_compressed(os.path.join(HERE, 'multiclass.npz'), **out)

    # ---- A5: theta-OBB codec ----------------------------------------------
    out = {}
    rois = torch.tensor([[0., 0., 9., 19.], [5., 5., 24., 14.]])
    deltas = torch.tensor([[.1, .2, .3, -.1, .05], [0., 0., 0., 0., 0.]])
    out['small_rois'] = rois.numpy()
    out['small_deltas'] = deltas.numpy()
    out['small_thetaobb'] = transforms.delta2thetaobb(rois, deltas).numpy()
    R = torch.from_numpy(rand_hbb(rng, 64, 512))
    D = torch.randn(64, 15) * 0.5
    stds = [.1, .1, .2, .2, .1]
    out['rois'] = R.numpy()
    out['deltas'] = D.numpy()
    out['stds'] = np.array(stds, dtype=np.float32)
    dec = transforms.delta2thetaobb(R, D, [0, 0, 0, 0, 0], stds)
    out['thetaobb'] = dec.numpy()
    gt = dec[:, 5:10].clone()
    out['gt'] = gt.numpy()
    out['enc'] = transforms.thetaobb2delta(R, gt, [0, 0, 0, 0, 0], stds).numpy()
    out['flip_in'] = dec.numpy()
    out['flip_out'] = transforms.thetaobb_flip(dec.numpy(), (600, 800, 3))
    out['rescale_out'] = transforms.thetaobb_rescale(dec.clone(), 1.7).numpy()
    out['rescale_rev_out'] = transforms.thetaobb_rescale(dec.clone(), 1.7, reverse_flag=True).numpy()
    np.savez_compressed(os.path.join(HERE, 'thetaobb_codec.npz'), **out)
    # ---- the other two head encodings of core/rbbox/transforms.py: pointobb (8 numbers) and hobb
    # (two corner points + height), incl. the host-side point-order helpers.  The cv2-based routines
    # (pointobb2thetaobb, thetaobb2pointobb, hobb_flip) cannot run here (no cv2 in the image).
    out = {}
    rng3 = np.random.default_rng(4242)
    torch.manual_seed(11)
    Rp = torch.from_numpy(rand_hbb(rng3, 48, 600))
    gtp = Rp[:, [0, 1, 2, 1, 2, 3, 0, 3]] + torch.randn(48, 8) * 6            # quads near the proposal corners
    m8, s8 = [0.01, -0.02, 0., 0.03, 0., 0., 0.02, -0.01], [.1, .1, .2, .2, .1, .1, .2, .2]
    out['p_rois'] = Rp.numpy(); out['p_gt'] = gtp.numpy()
    out['p_means'] = np.array(m8, np.float32); out['p_stds'] = np.array(s8, np.float32)
    out['p_enc'] = transforms.pointobb2delta(Rp, gtp, m8, s8).numpy()
    out['p_enc_default'] = transforms.pointobb2delta(Rp, gtp).numpy()
    Dp = torch.randn(48, 24) * 0.4
    out['p_deltas'] = Dp.numpy()
    out['p_dec'] = transforms.delta2pointobb(Rp, Dp, m8, s8).numpy()
    out['p_dec_default'] = transforms.delta2pointobb(Rp, Dp[:, :8]).numpy()
    gth = torch.cat([gtp[:, :4], (torch.rand(48, 1) * 80 + 4)], 1)
    m5, s5 = [0., 0.01, -0.01, 0., 0.02], [.1, .1, .2, .2, .1]
    out['h_gt'] = gth.numpy(); out['h_means'] = np.array(m5, np.float32); out['h_stds'] = np.array(s5, np.float32)
    out['h_enc'] = transforms.hobb2delta(Rp, gth, m5, s5).numpy()
    Dh = torch.randn(48, 15) * 0.6
    Dh[::7, 4::5] = 9.0                                                            # beyond the log-ratio clamp
    out['h_deltas'] = Dh.numpy()
    out['h_dec'] = transforms.delta2hobb(Rp, Dh, m5, s5).numpy()
    out['h_dec_default'] = transforms.delta2hobb(Rp, Dh[:, :5]).numpy()
    out['p_rescale'] = transforms.pointobb_rescale(gtp.clone(), 1.3).numpy()
    out['p_rescale_rev'] = transforms.pointobb_rescale(gtp.clone(), 1.3, reverse_flag=True).numpy()
    out['h_rescale'] = transforms.hobb_rescale(gth.clone(), 0.8).numpy()
    out['h_rescale_rev'] = transforms.hobb_rescale(gth.clone(), 0.8, reverse_flag=True).numpy()
    quads = gtp.numpy().astype(np.float64)
    quads[5] = [10., 20., 50., 20., 50., 60., 10., 60.]                           # axis-aligned: ties in both sort rules
    quads[6] = [50., 20., 50., 60., 10., 60., 10., 20.]
    out['q_in'] = quads
    out['q_bbox'] = np.array([transforms.pointobb2bbox(q.tolist()) for q in quads])
    out['q_extreme'] = np.array([transforms.pointobb_extreme_sort(q.tolist()) for q in quads])
    out['q_best'] = np.array([transforms.pointobb_best_point_sort(q.tolist()) for q in quads])
    out['q_flip'] = transforms.pointobb_flip(quads, (600, 800, 3))
    out['q_flip_one'] = transforms.pointobb_flip(quads[3], (600, 800, 3))
    out['q_map'] = transforms.pointobb_mapping(quads, (900, 1200, 3), 1.5, True)
    out['q_map_noflip'] = transforms.pointobb_mapping(quads, (900, 1200, 3), 1.5, False)
    out['q_mapback'] = transforms.pointobb_mapping_back(quads, (900, 1200, 3), 1.5, True)
    hb = np.concatenate([quads[:, :4], np.abs(rng3.normal(40, 15, (48, 1)))], 1)
    out['hq_in'] = hb
    out['hq_points'] = np.array([transforms.hobb2pointobb(h.tolist()) for h in hb])
    out['hq_map_noflip'] = transforms.hobb_mapping(hb, (900, 1200, 3), 1.5, False)
    out['hq_mapback_noflip'] = transforms.hobb_mapping_back(hb, (900, 1200, 3), 1.5, False)
    # ---- rbbox_target (core/rbbox/rbbox_target.py): training targets of the rbbox head, all encodings
    for name in ('mmdet.core', 'mmdet.core.rbbox', 'mmdet.core.utils'):
        if name not in sys.modules:
            m = types.ModuleType(name); m.__path__ = []; sys.modules[name] = m
    sys.modules['mmdet.core.rbbox.transforms'] = transforms
    sys.modules['mmdet.core.utils'].multi_apply = lambda fn, *args, **kw: tuple(map(list, zip(*map(lambda *a: fn(*a, **kw), *args))))
    rt = _load('mmdet.core.rbbox.rbbox_target', REF + '/mmdet/core/rbbox/rbbox_target.py')

    class _Enc(object):
        def __init__(self, e):
            self.encode = e
    pos = [Rp[:14], Rp[14:14], Rp[20:31]]                    # image 1 has no positives
    neg = [Rp[31:40], Rp[40:44], Rp[44:44]]                  # image 2 has no negatives
    inds = [torch.from_numpy(rng3.integers(0, 6, 14)), torch.zeros(0, dtype=torch.long), torch.from_numpy(rng3.integers(0, 3, 11))]
    glab = [torch.from_numpy(rng3.integers(1, 16, 6)), torch.from_numpy(rng3.integers(1, 16, 2)), torch.from_numpy(rng3.integers(1, 16, 3))]
    for k in range(3):
        out['t_pos_%d' % k] = pos[k].numpy(); out['t_neg_%d' % k] = neg[k].numpy()
        out['t_inds_%d' % k] = inds[k].numpy(); out['t_labels_%d' % k] = glab[k].numpy()
    gts = {'thetaobb': [torch.cat([gtp[:6, :2], gtp[:6, 2:4].abs() + 5, torch.rand(6, 1) - 1.2], 1),
                        torch.cat([gtp[6:8, :2], gtp[6:8, 2:4].abs() + 5, torch.rand(2, 1) - 1.2], 1),
                        torch.cat([gtp[8:11, :2], gtp[8:11, 2:4].abs() + 5, torch.rand(3, 1) - 1.2], 1)],
           'pointobb': [gtp[:6], gtp[6:8], gtp[8:11]],
           'hobb': [gth[:6], gth[6:8], gth[8:11]]}
    cfgt = {'thetaobb': ([0.] * 5, [.1, .1, .2, .2, .1], 5), 'pointobb': (m8, s8, 8), 'hobb': (m5, s5, 5)}
    import contextlib, io
    for enc, gl in gts.items():
        mm, ss, dim = cfgt[enc]
        for k in range(3):
            out['t_gt_%s_%d' % (enc, k)] = gl[k].numpy()
        with contextlib.redirect_stdout(io.StringIO()):      # ("num_pos = 0" is printed for image 1)
            res = rt.rbbox_target(pos, neg, inds, gl, glab, _Enc(enc), mm, ss, out_dim_reg=dim)
            per = rt.rbbox_target(pos, neg, inds, gl, glab, _Enc(enc), mm, ss, out_dim_reg=dim, concat=False)
        for nm, v in zip(('labels', 'label_weights', 'targets', 'weights'), res):
            out['t_%s_%s' % (enc, nm)] = v.numpy()
        out['t_%s_split' % enc] = np.array([len(x) for x in per[0]])
    np.savez_compressed(os.path.join(HERE, 'obb_codecs.npz'), **out)
    # ---- MaxIoUAssigner (mmdet/core/bbox/assigners/max_iou_assigner.py:8-202) -------
    out = {}
    pk = types.ModuleType('refbbox'); pk.__path__ = []; sys.modules['refbbox'] = pk
    sys.modules['refbbox.geometry'] = geometry
    pa = types.ModuleType('refbbox.assigners'); pa.__path__ = []; sys.modules['refbbox.assigners'] = pa
    adir = REF + '/mmdet/core/bbox/assigners/'
    mu = types.ModuleType('mmdet.utils'); mu.__path__ = []; sys.modules['mmdet.utils'] = mu
    mu.util_mixins = _load('mmdet.utils.util_mixins', REF + '/mmdet/utils/util_mixins.py')
    _load('refbbox.assigners.assign_result', adir + 'assign_result.py')
    _load('refbbox.assigners.base_assigner', adir + 'base_assigner.py')
    mia = _load('refbbox.assigners.max_iou_assigner', adir + 'max_iou_assigner.py')
    boxes = torch.from_numpy(rand_hbb(rng, 400, 300, 10, 80))
    gts = torch.from_numpy(rand_hbb(rng, 14, 300, 20, 90))
    ign = torch.from_numpy(rand_hbb(rng, 3, 300, 40, 120))
    labels = torch.from_numpy(rng.integers(1, 16, 14))
    for tag, kw, use_ign in (('a', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3), False),
                             ('b', dict(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.0,
                                        gt_max_assign_all=False), False),
                             ('c', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                        ignore_iof_thr=0.5), True)):
        a = mia.MaxIoUAssigner(**kw)
        r = a.assign(boxes, gts, ign if use_ign else None, labels)
        out['assign_%s_gt_inds' % tag] = r.gt_inds.numpy()
        out['assign_%s_max_overlaps' % tag] = r.max_overlaps.numpy()
        out['assign_%s_labels' % tag] = r.labels.numpy()
    out['boxes'] = boxes.numpy(); out['gts'] = gts.numpy(); out['ign'] = ign.numpy()
    out['labels'] = labels.numpy()
    out['overlaps'] = geometry.bbox_overlaps(gts, boxes).numpy()
    np.savez_compressed(os.path.join(HERE, 'assigner.npz'), **out)
    # ---- Soft-NMS (own generator: the fixtures above keep their random streams) -------
    rng2 = np.random.default_rng(20261004)
    out = {}
    doc6 = np.array([[4., 3., 5., 3., 0.9], [4., 3., 5., 4., 0.9], [3., 1., 3., 1., 0.5],
                     [3., 1., 3., 1., 0.5], [3., 1., 3., 1., 0.4], [3., 1., 3., 1., 0.0]], dtype=np.float32)
    out['doc6_dets'] = doc6
    nd, inds = nms_wrapper.soft_nms(doc6, 0.7, sigma=0.5)
    out['doc6_new_dets'] = nd; out['doc6_inds'] = inds
    t4 = np.array([[49.1, 32.4, 51.0, 35.9, 0.9], [49.3, 32.9, 51.0, 35.3, 0.9],
                   [35.3, 11.5, 39.9, 14.5, 0.4], [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
    out['test4_dets'] = t4
    nd, inds = nms_wrapper.soft_nms(t4, 0.7)
    out['test4_new_dets'] = nd; out['test4_inds'] = inds
    for tag, n, canvas, thr, method, sigma, min_score, ties in (
            ('a', 300, 128, 0.3, 'linear', 0.5, 1e-3, False),
            ('b', 300, 128, 0.3, 'gaussian', 0.5, 1e-3, False),
            ('c', 2000, 400, 0.5, 'linear', 0.5, 0.05, True),
            ('d', 2000, 400, 0.5, 'gaussian', 0.3, 0.05, True),
            ('e', 1000, 1000, 0.3, 'linear', 0.5, 1e-3, False)):
        boxes = rand_hbb(rng2, n, canvas)
        score = rng2.uniform(0, 1, n).astype(np.float32)
        if ties:
            score[::3] = np.float32(0.5)       # equal scores: the first position wins (:114-121)
        dets = np.concatenate([boxes, score[:, None]], 1)
        nd, inds = nms_wrapper.soft_nms(dets, thr, method=method, sigma=sigma, min_score=min_score)
        out['rand_%s_dets' % tag] = dets
        out['rand_%s_cfg' % tag] = np.array([thr, sigma, min_score], dtype=np.float64)
        out['rand_%s_method' % tag] = np.array(method)
        out['rand_%s_new_dets' % tag] = nd
        out['rand_%s_inds' % tag] = inds
    np.savez_compressed(os.path.join(HERE, 'soft_nms.npz'), **out)

    # ---- HBB NMS on integer boxes: exact IoU == thr ties (nms_cpu.cpp:56 `>=`) ----------
    out = {}
    two = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 19, 0.8], [0, 0, 9, 29, 0.7]], dtype=np.float32)
    out['half_dets'] = two                      # IoU(0,1) = 100/200 = 0.5 exactly
    out['half_keep'] = nms_wrapper.nms(two, 0.5)[1]
    for tag, n, canvas, thr in (('a', 2000, 96, 0.5), ('b', 2000, 64, 0.25), ('c', 3000, 200, 0.5)):
        boxes = rand_hbb(rng2, n, canvas, 4, 40, integer=True)
        score = ((rng2.permutation(n) + 0.5) / n).astype(np.float32)
        dets = np.concatenate([boxes, score[:, None]], 1)
        out['int_%s_dets' % tag] = dets
        out['int_%s_thr' % tag] = np.float32(thr)
        out['int_%s_keep' % tag] = nms_wrapper.nms(dets, thr)[1]
    np.savez_compressed(os.path.join(HERE, 'hbb_nms_ties.npz'), **out)
    # ---- class-wise orchestration with type='soft_nms' (rbbox_nms.py:25-43 resolves any op by name)
    g = np.load(os.path.join(HERE, 'multiclass.npz'))
    mb, ms = torch.from_numpy(g['multi_bboxes']), torch.from_numpy(g['multi_scores'])
    cfg = dict(type='soft_nms', iou_thr=0.5, min_score=0.05)
    bb, ll, ci, ki = rbbox_nms.multiclass_nms_with_index(mb, ms, 0.05, cfg, 100)
    out = {'idx_bboxes': bb.numpy(), 'idx_labels': ll.numpy()}
    for i, k in enumerate(ki):
        out['idx_keep_%d' % i] = k.numpy()
    m2, l2 = bbox_nms.multiclass_nms(mb, ms, 0.05, cfg, 100)
    out['mc_bboxes'] = m2.numpy(); out['mc_labels'] = l2.numpy()
    np.savez_compressed(os.path.join(HERE, 'multiclass_soft.npz'), **out)
    detector_parts(nms_wrapper, bbox_nms)
    print('wrote', sorted(f for f in os.listdir(HERE) if f.endswith('.npz')))


def detector_parts(nms_wrapper, bbox_nms):
    """Load the reference's RPNHead / AnchorHead / SingleRoIExtractor / BBoxHead classes with stub
    packages around them (mmcv, registries and losses are not needed on the test path) and
    capture their outputs on seeded random network outputs."""
    from torch import nn

    def pkg(name):
        m = sys.modules.get(name)
        if m is None:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
        return m

    class _Registry(object):
        def register_module(self, cls):
            return cls

    def force_fp32(*a, **k):
        return lambda fn: fn

    def auto_fp16(*a, **k):
        return lambda fn: fn

    anchor_gen = _load('ref_anchor_generator', REF + '/mmdet/core/anchor/anchor_generator.py')
    pkg('mmcv'); cnn = pkg('mmcv.cnn'); cnn.normal_init = lambda *a, **k: None
    sys.modules['mmcv'].cnn = cnn
    bt = _load('ref_bbox_transforms', REF + '/mmdet/core/bbox/transforms.py')
    core = pkg('mmdet.core')
    core.AnchorGenerator = anchor_gen.AnchorGenerator
    core.delta2bbox = bt.delta2bbox
    core.force_fp32 = force_fp32
    core.auto_fp16 = auto_fp16
    core.multi_apply = lambda fn, *args: tuple(map(list, zip(*map(fn, *args))))
    core.anchor_target = None
    core.multiclass_nms = bbox_nms.multiclass_nms
    core.bbox_target = None
    ops = pkg('mmdet.ops'); ops.nms = nms_wrapper.nms
    sys.modules['mmdet'].ops = ops
    models = pkg('mmdet.models')
    reg = pkg('mmdet.models.registry'); reg.HEADS = _Registry(); reg.ROI_EXTRACTORS = _Registry()
    bld = pkg('mmdet.models.builder'); bld.build_loss = lambda cfg: nn.Module()
    loss = pkg('mmdet.models.losses'); loss.accuracy = None
    pkg('mmdet.models.anchor_heads'); pkg('mmdet.models.roi_extractors'); pkg('mmdet.models.bbox_heads')
    _load('mmdet.models.anchor_heads.anchor_head', REF + '/mmdet/models/anchor_heads/anchor_head.py')
    rpn = _load('mmdet.models.anchor_heads.rpn_head', REF + '/mmdet/models/anchor_heads/rpn_head.py')
    sl = _load('mmdet.models.roi_extractors.single_level', REF + '/mmdet/models/roi_extractors/single_level.py')
    bh = _load('mmdet.models.bbox_heads.bbox_head', REF + '/mmdet/models/bbox_heads/bbox_head.py')

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    rng = np.random.default_rng(20261005)
    torch.manual_seed(5)
    out = {}
    # ---- anchors: config 4's generators (strides 4..64, scale 8, ratios .5/1/2)
    strides = [4, 8, 16, 32, 64]
    gens = [anchor_gen.AnchorGenerator(s, [8], [0.5, 1.0, 2.0]) for s in strides]
    out['anchor_strides'] = np.array(strides)
    for i, g in enumerate(gens):
        out['base_anchors_%d' % i] = g.base_anchors.numpy()
        out['grid_small_%d' % i] = g.grid_anchors((3, 5), strides[i], device='cpu').numpy()
        full = g.grid_anchors((1024 // strides[i], 1024 // strides[i]), strides[i], device='cpu').numpy()
        out['grid_tile_rows_%d' % i] = np.arange(0, len(full), 997)
        out['grid_tile_sample_%d' % i] = full[::997]
        out['grid_tile_sum_%d' % i] = full.astype(np.float64).sum(0)
        out['grid_tile_len_%d' % i] = np.int64(len(full))
    g9 = anchor_gen.AnchorGenerator(9, [1.], [1.])
    out['doc_grid'] = g9.grid_anchors((2, 2), device='cpu').numpy()
    g2 = anchor_gen.AnchorGenerator(16, [2., 4.], [0.5, 2.0], scale_major=False, ctr=(3.0, 5.0))
    out['ctr_base_anchors'] = g2.base_anchors.numpy()
    # ---- delta2bbox
    rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    out['d2b_doc_rois'] = rois.numpy(); out['d2b_doc_deltas'] = deltas.numpy()
    out['d2b_doc_out'] = bt.delta2bbox(rois, deltas, max_shape=(32, 32)).numpy()
    R = torch.from_numpy(rand_hbb(rng, 300, 900))
    D = torch.randn(300, 16 * 4) * 0.7
    out['d2b_rois'] = R.numpy(); out['d2b_deltas'] = D.numpy()
    out['d2b_out'] = bt.delta2bbox(R, D, [0., 0., 0., 0.], [0.1, 0.1, 0.2, 0.2], (1024, 1024, 3)).numpy()
    out['d2b_out_noclip'] = bt.delta2bbox(R, D[:, :4], [0.1, -0.1, 0.05, 0.], [1., 1., 1., 1.]).numpy()
    out['bbox2roi'] = bt.bbox2roi([R[:3, :], R[:0, :], torch.cat([R[3:5], torch.ones(2, 1)], 1)]).numpy()
    # ---- RPNHead.get_bboxes (anchor_head.py:254-283 -> rpn_head.py:55-104) on a 256 x 256 image
    head = rpn.RPNHead(in_channels=8, feat_channels=8, anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                       anchor_strides=strides, target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
    sizes = [64, 32, 16, 8, 4]
    cls = [torch.randn(1, 3, s, s) * 2 for s in sizes]
    reg = [torch.randn(1, 12, s, s) * 0.5 for s in sizes]
    for i in range(5):
        out['rpn_cls_%d' % i] = cls[i].numpy(); out['rpn_reg_%d' % i] = reg[i].numpy()
    metas = [dict(img_shape=(256, 256, 3), scale_factor=1.0)]
    cfgs = {'a': Cfg(nms_across_levels=False, nms_pre=300, nms_post=300, max_num=300, nms_thr=0.7, min_bbox_size=0),
            'b': Cfg(nms_across_levels=False, nms_pre=500, nms_post=120, max_num=200, nms_thr=0.5, min_bbox_size=8),
            'c': Cfg(nms_across_levels=True, nms_pre=200, nms_post=200, max_num=150, nms_thr=0.7, min_bbox_size=0)}
    for tag, cfg in cfgs.items():
        props = head.get_bboxes(cls, reg, metas, cfg)[0]
        out['rpn_%s_cfg' % tag] = np.array([cfg.nms_across_levels, cfg.nms_pre, cfg.nms_post, cfg.max_num,
                                             cfg.nms_thr, cfg.min_bbox_size], dtype=np.float64)
        out['rpn_%s_proposals' % tag] = props.numpy()
    # ---- SingleRoIExtractor.map_roi_levels / roi_rescale
    ext = sl.SingleRoIExtractor.__new__(sl.SingleRoIExtractor)
    ext.finest_scale = 56
    rr = torch.cat([torch.zeros(400, 1), torch.from_numpy(rand_hbb(rng, 400, 900, 4, 700))], 1)
    rr[:8, 1:] = torch.tensor([[0, 0, 111, 111], [0, 0, 110, 111], [0, 0, 223, 223], [0, 0, 222, 223],
                               [0, 0, 447, 447], [0, 0, 446, 447], [0, 0, 55, 55], [0, 0, 2000, 2000.]])
    out['lvl_rois'] = rr.numpy()
    out['lvl_target'] = ext.map_roi_levels(rr, 4).numpy()
    out['lvl_rescaled'] = ext.roi_rescale(rr, 1.25).numpy()
    # ---- BBoxHead.get_det_bboxes without NMS (softmax + decode + rescale)
    bbh = bh.BBoxHead.__new__(bh.BBoxHead)
    bbh.target_means = [0., 0., 0., 0.]; bbh.target_stds = [0.1, 0.1, 0.2, 0.2]
    cs = torch.randn(300, 16) * 2
    roi5 = torch.cat([torch.zeros(300, 1), R], 1)
    bb, sc = bbh.get_det_bboxes(roi5, cs, D, (1024, 1024, 3), 2.0, rescale=True, cfg=None)
    out['head_cls_score'] = cs.numpy(); out['head_bboxes'] = bb.numpy(); out['head_scores'] = sc.numpy()
    # ---- test-time augmentation merges (core/post_processing/merge_augs.py, core/bbox/transforms.py:114-146)
    pkg('mmdet.core.bbox').bbox_mapping_back = bt.bbox_mapping_back
    pkg('mmdet.core.post_processing')
    ma = _load('mmdet.core.post_processing.merge_augs', REF + '/mmdet/core/post_processing/merge_augs.py')
    rng2 = np.random.default_rng(777)
    fb = torch.from_numpy(rand_hbb(rng2, 40, 380).reshape(20, 8))
    out['aug_flip_in'] = fb.numpy()
    out['aug_flip_out'] = bt.bbox_flip(fb, (300, 400, 3)).numpy()
    out['aug_map_out'] = bt.bbox_mapping(fb, (450, 600, 3), 1.5, True).numpy()
    out['aug_map_noflip'] = bt.bbox_mapping(fb, (450, 600, 3), 1.5, False).numpy()
    out['aug_mapback_out'] = bt.bbox_mapping_back(fb, (450, 600, 3), 1.5, True).numpy()
    metas = [dict(img_shape=(300, 400, 3), scale_factor=1.0, flip=False),
             dict(img_shape=(450, 600, 3), scale_factor=1.5, flip=True),
             dict(img_shape=(225, 300, 3), scale_factor=0.75, flip=False)]
    props = []
    for k, (m, cnt) in enumerate(zip(metas, (150, 130, 90))):
        b = rand_hbb(rng2, cnt, int(380 * m['scale_factor']), 8, int(110 * m['scale_factor']))
        sc = (rng2.permutation(1000)[:cnt] * 3 + k + 0.5) / 3000.0      # distinct over all three lists
        props.append(torch.from_numpy(np.concatenate([b, sc[:, None]], 1).astype(np.float32)))
        out['aug_props_%d' % k] = props[-1].numpy()
    out['aug_metas'] = np.array([[m['img_shape'][0], m['img_shape'][1], m['scale_factor'], float(m['flip'])] for m in metas])
    out['aug_props_merged'] = ma.merge_aug_proposals(props, metas, Cfg(nms_thr=0.7, max_num=120)).numpy()
    out['aug_props_merged_all'] = ma.merge_aug_proposals(props, metas, Cfg(nms_thr=0.5, max_num=100000)).numpy()
    ab = [torch.from_numpy(rand_hbb(rng2, 50 * 16, 380).reshape(50, 64)) * m['scale_factor'] for m in metas]
    asc = [torch.from_numpy(rng2.random((50, 16)).astype(np.float32)) for _ in metas]
    for k in range(3):
        out['aug_bboxes_%d' % k] = ab[k].numpy(); out['aug_scores_%d' % k] = asc[k].numpy()
    mb, ms = ma.merge_aug_bboxes(ab, asc, [[m] for m in metas], None)
    out['aug_bboxes_merged'] = mb.numpy(); out['aug_scores_merged'] = ms.numpy()
    out['aug_bboxes_only'] = ma.merge_aug_bboxes(ab, None, [[m] for m in metas], None).numpy()
    out['aug_scores_mean_np'] = ma.merge_aug_scores([a.numpy() for a in asc])
    out['aug_scores_mean_t'] = ma.merge_aug_scores(asc).numpy()
    np.savez_compressed(os.path.join(HERE, 'detector_parts.npz'), **out)


if __name__ == '__main__':
    main()
