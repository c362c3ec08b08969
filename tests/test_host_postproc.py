"""CPU tests of the host-side orchestration (aidet_amd.post_processing, assigners) with the
NMS op replaced by the oracle, against the fixtures captured from the reference, and of the
oracle's newer restatements (Soft-NMS, exact-tie HBB NMS, strict compare, polyiou fp64)
against their fixtures.  No GPU needed: what is checked here is the host logic and the
checker itself; the HIP ops are checked against the same fixtures in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import oracle


def _oracle_op(fn):
    """An nms_wrapper-style op `(dets, iou_thr, groups=None) -> (dets[keep], keep)` on the oracle."""
    def op(dets, iou_thr, device_id=None, groups=None, thrs=None):
        d = dets.detach().cpu().numpy()
        g = None if groups is None else groups.detach().cpu().numpy()
        kw = {} if thrs is None else {'thrs': thrs.detach().cpu().numpy()}
        keep = torch.from_numpy(fn(d, float(iou_thr), groups=g, **kw))
        return dets[keep], keep
    return op


@pytest.fixture
def oracle_ops(monkeypatch):
    from aidet_amd import nms_wrapper
    monkeypatch.setattr(nms_wrapper, 'nms', _oracle_op(oracle.hbb_nms))
    monkeypatch.setattr(nms_wrapper, 'rnms', _oracle_op(oracle.rnms))
    return nms_wrapper


def test_multiclass_nms_with_index_reproduces_reference(golden, oracle_ops):
    """rbbox_nms.py:6-119 captured I/O (anchor A3): per-class keep lists, class-major
    concatenation, top-k, and the keep re-use on the rotated boxes."""
    from aidet_amd.post_processing import multiclass_nms_with_index, thetaobb_nms_by_bbox_nms
    g = golden('multiclass.npz')
    mb, ms, mr = (torch.from_numpy(g[k]) for k in ('multi_bboxes', 'multi_scores', 'multi_rbboxes'))
    cfg = dict(type='nms', iou_thr=float(g['iou_thr']))
    bboxes, labels, cls_inds, keep_inds = multiclass_nms_with_index(mb, ms, float(g['score_thr']), cfg,
                                                                   int(g['max_num']))
    assert len(cls_inds) == ms.shape[1] - 1 and all(c.dtype == torch.bool for c in cls_inds)
    assert [len(k) for k in keep_inds] == g['idx_keep_lens'].tolist()
    for i, k in enumerate(keep_inds):
        assert k.dtype == torch.int64 and np.array_equal(k.numpy(), g['idx_keep_%d' % i])
    assert np.array_equal(bboxes.numpy(), g['idx_bboxes'])
    assert np.array_equal(labels.numpy(), g['idx_labels'])
    n_lists = len(keep_inds)
    rb, rl = thetaobb_nms_by_bbox_nms(mr, ms, cls_inds, keep_inds, int(g['max_num']), 5)
    assert np.array_equal(rb.numpy(), g['rb_bboxes'])
    assert np.array_equal(rl.numpy(), g['rb_labels'])
    assert n_lists > 0 and len(keep_inds) == 0       # consumed like the reference's pop(0)


def test_multiclass_nms_reproduces_reference(golden, oracle_ops):
    """bbox_nms.py:6-76 captured I/O (mc_bboxes / mc_labels): the coordinate-offset form."""
    from aidet_amd.post_processing import multiclass_nms
    g = golden('multiclass.npz')
    mb, ms = torch.from_numpy(g['multi_bboxes']), torch.from_numpy(g['multi_scores'])
    cfg = dict(type='nms', iou_thr=float(g['iou_thr']))
    det, lab = multiclass_nms(mb, ms, float(g['score_thr']), cfg, int(g['max_num']))
    assert np.array_equal(det.numpy(), g['mc_bboxes'])
    assert np.array_equal(lab.numpy(), g['mc_labels'])
    # the grouped form decides the same on this input (no offset rounding on a threshold)
    det2, lab2 = multiclass_nms(mb, ms, float(g['score_thr']), cfg, int(g['max_num']), grouped=True)
    assert np.array_equal(det2.numpy(), g['mc_bboxes']) and np.array_equal(lab2.numpy(), g['mc_labels'])


def test_multiclass_with_soft_nms_reproduces_reference(golden, monkeypatch):
    """nms_cfg type='soft_nms' (the alternative the config's comment names, dota_v002...:133-134):
    class by class like the reference; keep lists, re-scored boxes and the offset-trick variant."""
    from aidet_amd import nms_wrapper
    from aidet_amd.post_processing import multiclass_nms, multiclass_nms_with_index

    def soft(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3, device_id=None):
        nd, inds = oracle.soft_nms(dets.numpy(), iou_thr, method, sigma, min_score)
        return torch.from_numpy(nd), torch.from_numpy(inds)
    monkeypatch.setattr(nms_wrapper, 'soft_nms', soft)
    g, w = golden('multiclass.npz'), golden('multiclass_soft.npz')
    mb, ms = torch.from_numpy(g['multi_bboxes']), torch.from_numpy(g['multi_scores'])
    cfg = dict(type='soft_nms', iou_thr=0.5, min_score=0.05)
    b, l, ci, ki = multiclass_nms_with_index(mb, ms, 0.05, cfg, 100)
    assert np.array_equal(b.numpy(), w['idx_bboxes']) and np.array_equal(l.numpy(), w['idx_labels'])
    for i, k in enumerate(ki):
        assert np.array_equal(k.numpy(), w['idx_keep_%d' % i])
    d, l2 = multiclass_nms(mb, ms, 0.05, cfg, 100)
    assert np.array_equal(d.numpy(), w['mc_bboxes']) and np.array_equal(l2.numpy(), w['mc_labels'])


def test_multiclass_edge_cases(oracle_ops):
    from aidet_amd.post_processing import (multiclass_nms, multiclass_nms_with_index, multiclass_rnms,
                                           thetaobb_nms_by_bbox_nms)
    ms = torch.tensor([[0.9, 0.01, 0.02], [0.8, 0.03, 0.01]])
    mb = torch.rand(2, 12) * 50
    cfg = dict(type='nms', iou_thr=0.5)
    # nothing passes the score threshold
    b, l, ci, ki = multiclass_nms_with_index(mb, ms, 0.05, cfg, 100)
    assert b.shape == (0, 5) and l.shape == (0,) and l.dtype == torch.long and len(ci) == 2 and ki == []
    rb, rl = thetaobb_nms_by_bbox_nms(torch.rand(2, 15), ms, ci, ki, 100, 5)
    assert rb.shape == (0, 6) and rl.shape == (0,)
    d, l = multiclass_nms(mb, ms, 0.05, cfg, 100)
    assert d.shape == (0, 5) and l.dtype == torch.long
    d, l = multiclass_rnms(torch.rand(2, 15), ms, 0.05, dict(type='rnms', iou_thr=0.5), 100)
    assert d.shape == (0, 6)
    # the positional signature of the reference's disabled call (rbbox_head.py:294-295)
    from aidet_amd import multiclass_thetaobb_nms
    d, l = multiclass_thetaobb_nms(torch.rand(2, 15), ms, 0.05, 0.5, 100)
    assert d.shape == (0, 6) and l.dtype == torch.long
    # one class empty, shared (n, 4) boxes, max_num = -1 quirk of the reference (inds[:-1])
    ms = torch.tensor([[0.1, 0.6, 0.0], [0.1, 0.7, 0.0], [0.1, 0.8, 0.0]])
    mb4 = torch.tensor([[0., 0, 10, 10], [100, 100, 120, 120], [200, 200, 230, 230]])
    b, l, ci, ki = multiclass_nms_with_index(mb4, ms, 0.05, cfg, -1)
    assert len(ki) == 1 and ki[0].tolist() == [0, 1, 2]
    assert b[:, 4].tolist() == pytest.approx([0.8, 0.7]) and l.tolist() == [0, 0]
    b, l, _, _ = multiclass_nms_with_index(mb4, ms, 0.05, cfg, 100)
    assert b[:, 4].tolist() == pytest.approx([0.6, 0.7, 0.8])     # not re-ordered below the cap


def test_assign_wrt_overlaps_reproduces_reference(golden):
    from aidet_amd.assigners import assign_wrt_overlaps
    g = golden('assigner.npz')
    ov = torch.from_numpy(g['overlaps'])
    labels = torch.from_numpy(g['labels'])
    for tag, kw in (('a', dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3)),
                    ('b', dict(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.0,
                               gt_max_assign_all=False))):
        r = assign_wrt_overlaps(ov.clone(), labels, **kw)
        assert np.array_equal(r.gt_inds.numpy(), g['assign_%s_gt_inds' % tag]), tag
        assert np.array_equal(r.labels.numpy(), g['assign_%s_labels' % tag]), tag
        assert np.array_equal(r.max_overlaps.numpy(), g['assign_%s_max_overlaps' % tag]), tag
    # two gts claiming the same box: the LATER gt wins (the reference's loop order)
    ov = torch.tensor([[0.6, 0.2, 0.0], [0.6, 0.1, 0.3]])
    r = assign_wrt_overlaps(ov, None, pos_iou_thr=0.9, neg_iou_thr=0.1, min_pos_iou=0.0)
    assert r.gt_inds.tolist() == [2, -1, -1]
    r = assign_wrt_overlaps(ov, None, pos_iou_thr=0.9, neg_iou_thr=0.1, min_pos_iou=0.0, gt_max_assign_all=False)
    assert r.gt_inds.tolist() == [2, -1, -1]
    r = assign_wrt_overlaps(ov, None, pos_iou_thr=0.9, neg_iou_thr=0.25, min_pos_iou=0.65)
    assert r.gt_inds.tolist() == [-1, 0, -1]


# ---- oracle restatements against their fixtures -------------------------------------------
def test_soft_nms_oracle_reproduces_reference(golden):
    g = golden('soft_nms.npz')
    nd, inds = oracle.soft_nms(g['doc6_dets'], 0.7, sigma=0.5)
    assert len(inds) == 3                                            # nms_wrapper.py:80-89 doctest
    assert np.array_equal(inds, g['doc6_inds']) and np.array_equal(nd, g['doc6_new_dets'])
    nd, inds = oracle.soft_nms(g['test4_dets'], 0.7)
    assert len(inds) == 4                                            # tests/test_soft_nms.py
    assert np.array_equal(inds, g['test4_inds']) and np.array_equal(nd, g['test4_new_dets'])
    for tag in 'abcde':
        thr, sigma, min_score = g['rand_%s_cfg' % tag]
        method = str(g['rand_%s_method' % tag])
        nd, inds = oracle.soft_nms(g['rand_%s_dets' % tag], thr, method, sigma, min_score)
        assert np.array_equal(inds, g['rand_%s_inds' % tag]), tag
        assert np.array_equal(nd[:, :4], g['rand_%s_new_dets' % tag][:, :4]), tag
        assert np.array_equal(nd[:, 4], g['rand_%s_new_dets' % tag][:, 4]), tag      # (gaussian: the reference's std::exp(float) and the oracle's expf are the same glibc routine)
    with pytest.raises(ValueError):
        oracle.soft_nms(g['doc6_dets'], 0.7, method='bogus')


def test_hbb_nms_exact_ties(golden):
    """Integer boxes give IoU == thr exactly; nms_cpu.cpp:56 suppresses on `>=`."""
    g = golden('hbb_nms_ties.npz')
    assert g['half_keep'].tolist() == [0, 2]
    assert oracle.hbb_nms(g['half_dets'], 0.5).tolist() == [0, 2]
    assert oracle.hbb_nms(g['half_dets'], 0.5, strict=True).tolist() == [0, 1]      # nms_kernel.cu:61 `>`: 1 survives and takes 2
    for tag in 'abc':
        d, thr, keep = g['int_%s_dets' % tag], float(g['int_%s_thr' % tag]), g['int_%s_keep' % tag]
        assert np.array_equal(oracle.hbb_nms(d, thr), keep), tag
        assert np.array_equal(oracle.hbb_nms(d, thr, groups=np.zeros(len(d), np.int32)), keep), tag
        strict = oracle.hbb_nms(d, thr, strict=True)
        assert len(strict) > len(keep), tag          # the ties are real: `>` keeps more


def test_hbb_nms_groups_equal_per_group_calls():
    rng = np.random.default_rng(11)
    n = 1500
    x1, y1 = rng.uniform(0, 200, n), rng.uniform(0, 200, n)
    d = np.stack([x1, y1, x1 + rng.uniform(5, 60, n), y1 + rng.uniform(5, 60, n),
                  (rng.permutation(n) + 0.5) / n], 1).astype(np.float32)
    grp = rng.integers(0, 7, n).astype(np.int32)
    got = oracle.hbb_nms(d, 0.4, groups=grp)
    want = np.sort(np.concatenate([np.nonzero(grp == c)[0][oracle.hbb_nms(d[grp == c], 0.4)] for c in range(7)]))
    assert np.array_equal(got, want)


def test_rotated_strict_compare_differs_only_on_ties():
    d = oracle.synth_boxes(3000, 400, 21)
    assert np.array_equal(oracle.rnms(d, 0.5), oracle.rnms(d, 0.5, strict=True))   # no exact ties at random
    two = np.array([[10, 10, 8, 4, 0.3, 0.9], [10, 10, 8, 4, 0.3, 0.8]], np.float32)
    assert oracle.rnms(two, 1.0).tolist() == [0] and oracle.rnms(two, 1.0, strict=True).tolist() == [0, 1]


def test_polyiou_third_formulation_agrees_with_the_spec():
    """>= 10^6 aligned pairs at DOTA coordinates (random + near-degenerate): the fp32 spec vs the
    DOTA devkit's triangle-fan algorithm restated in fp64 -- an independent formulation."""
    rng = np.random.default_rng(2026)
    n = 1 << 20
    a = oracle.synth_boxes(n, 20000, 1)[:, :5]
    # partner: a perturbed copy (so that most pairs overlap), several regimes
    b = a.copy()
    reg = rng.integers(0, 5, n)
    b[:, 0] += rng.normal(0, 1, n) * a[:, 2] * np.where(reg == 0, 0.5, np.where(reg == 1, 0.05, 1e-3))
    b[:, 1] += rng.normal(0, 1, n) * a[:, 3] * np.where(reg == 0, 0.5, np.where(reg == 1, 0.05, 1e-3))
    b[:, 2] *= np.exp(rng.normal(0, 1, n) * np.where(reg <= 1, 0.3, 1e-3))
    b[:, 3] *= np.exp(rng.normal(0, 1, n) * np.where(reg <= 1, 0.3, 1e-3))
    b[:, 4] += rng.normal(0, 1, n) * np.where(reg == 0, 0.6, np.where(reg == 3, 1e-4, 0.05))
    b[reg == 4] = a[reg == 4]                     # exact duplicates
    b[reg == 4, 4] += np.float32(np.pi / 2) * rng.integers(0, 2, (reg == 4).sum())   # ... or rotated by 90 deg
    b = b.astype(np.float32)
    got = oracle.overlaps(a, b, is_aligned=True).astype(np.float64)
    ref = oracle.polyiou_f64(a, b, recentre=True)
    diff = np.abs(got - ref)
    assert (ref > 0.01).mean() > 0.6              # the pairs do overlap
    # fp32 theta quantisation at |theta| ~ 1 moves a corner of a 128-px box by ~4e-6 px: the
    # spec's error against an fp64 formulation stays far inside the north star's 1e-4
    assert diff.max() <= 5e-6, diff.max()      # measured: 2.2e-6 (mean 9e-8)
    # the published form (fan from the coordinate origin) loses precision at DOTA coordinates:
    sub = slice(0, 1 << 16)
    pub = oracle.polyiou_f64(a[sub], b[sub], recentre=False)
    assert np.abs(pub - ref[sub]).max() <= 1e-6
    # decisions at the thresholds the merge uses (dota.py:324): a disagreement needs |iou - thr|
    # below the spec's error, so it is confined to a sliver of pairs
    for thr in (0.05, 0.1, 0.3, 0.5):
        flips = (got >= thr) != (ref >= thr)
        assert flips.sum() <= 1e-4 * n
        assert np.all(np.abs(ref[flips] - thr) <= 2e-5)
