"""GPU tests of the round-3 additions (run on the MI355X box: `pytest -m gpu`): the gradient and the row-block
partition of rbbox_overlaps, tickets collected from another stream, the workspace forward of RoI-Align (Morton-ordered
block numbering) against the plain entry points and the oracle."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device (no CPU path exists)'
    return torch.device('cuda:0')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def same_rows_up_to_score_ties(got, want, tol=2e-4):
    """Two (n, d + 1) detection lists in descending score order hold the same rows, rows of (nearly) equal score in any
    order; rows whose score equals the LAST score (a tie cut by the list's end) are compared by count only."""
    a, b = got.detach().cpu().numpy(), want.detach().cpu().numpy()
    if a.shape != b.shape or not np.allclose(a[:, -1], b[:, -1], atol=1e-6):
        return False
    if len(a) == 0:
        return True
    inner = a[:, -1] > a[-1, -1] + 1e-6

    def canon(x):
        x = x[inner]
        return x[np.lexsort((x[:, 1], x[:, 0], np.round(x[:, -1], 5)))]
    ca, cb = canon(a), canon(b)
    if ca.shape != cb.shape:
        return False
    # rows may still be permuted inside one rounding bucket of the score: match them greedily inside +-4 positions
    bad = np.abs(ca - cb).max(axis=1) > tol
    for i in np.nonzero(bad)[0]:
        lo, hi = max(0, i - 4), min(len(cb), i + 5)
        if not (np.abs(cb[lo:hi] - ca[i]).max(axis=1) <= tol).any():
            return False
    return True


def test_rbbox_overlaps_has_a_gradient_for_a_rotated_iou_loss():
    """iou_loss.py:25 differentiates through bbox_overlaps(pred, target, is_aligned=True); the rotated op does the same:
    forward = the kernel's values (bit-equal to the no-grad call), backward = autograd of the torch restatement."""
    from aidet_amd import rbbox_overlaps
    from aidet_amd.rbbox_geometry import rbbox_overlaps_torch
    a = oracle.synth_boxes(600, 200, 1)[:, :5]
    b = a + np.array([3., 2., 1., -1., 0.2], np.float32)
    pa, pb = t(a).requires_grad_(True), t(b)
    iou = rbbox_overlaps(pa, pb, is_aligned=True)
    assert iou.requires_grad and torch.equal(iou.detach(), rbbox_overlaps(t(a), t(b), is_aligned=True))
    (1 - iou).sum().backward()
    ref_in = torch.from_numpy(a).double().requires_grad_(True)
    (1 - rbbox_overlaps_torch(ref_in, torch.from_numpy(b).double(), 'iou', True)).sum().backward()
    assert pa.grad.shape == (600, 5) and float((pa.grad.cpu().double() - ref_in.grad).abs().max()) < 1e-3
    assert float(pa.grad.abs().max()) > 1e-3
    # all-pairs mode and the second argument
    qa, qb = t(a[:40]).requires_grad_(True), t(b[:50]).requires_grad_(True)
    rbbox_overlaps(qa, qb).sum().backward()
    assert qa.grad.shape == (40, 5) and qb.grad.shape == (50, 5) and float(qb.grad.abs().max()) > 0
    with pytest.raises(RuntimeError):      # a gradient over millions of pairs is refused, not silently dropped
        rbbox_overlaps(t(oracle.synth_boxes(2000, 500, 3)[:, :5]).requires_grad_(True), t(oracle.synth_boxes(2000, 500, 4)[:, :5]))


def test_rbbox_overlaps_row_shard_blocks_concatenate_to_the_full_matrix():
    """the multi-GPU partition of rbbox_overlaps is by row blocks, no communication."""
    from aidet_amd import rbbox_overlaps
    a = t(oracle.synth_boxes(1001, 600, 5)[:, :5])
    b = t(oracle.synth_boxes(777, 600, 6)[:, :5])
    full = rbbox_overlaps(a, b)
    for world in (1, 2, 3, 8):
        parts = [rbbox_overlaps(a, b, row_shard=(r, world)) for r in range(world)]
        assert torch.equal(torch.cat(parts, 0), full)
    al = rbbox_overlaps(a[:777], b, is_aligned=True)
    assert torch.equal(torch.cat([rbbox_overlaps(a[:777], b, is_aligned=True, row_shard=(r, 4)) for r in range(4)]), al)
    assert rbbox_overlaps(a[:3], b, row_shard=(7, 8)).shape == (0, 777)


def test_nms_tickets_enqueued_on_side_streams_are_collected_from_the_default_stream():
    """the ticket records an event on ITS stream; nms_collect waits for it before reading the record."""
    from aidet_amd import nms_wrapper
    scenes = [oracle.synth_boxes(3000 + 500 * i, 900, 20 + i) for i in range(5)]
    dets = [t(s) for s in scenes]
    streams = [torch.cuda.Stream() for _ in scenes]
    for rep in range(20):
        tickets = []
        for d, s in zip(dets, streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                # a long-running kernel in front of the NMS on the side stream: the collector must not overtake it
                junk = torch.randn(2048, 2048, device=dev()) @ torch.randn(2048, 2048, device=dev())
                tickets.append(nms_wrapper.nms_enqueue(d, 0.5, op='rnms'))
        keeps = nms_wrapper.nms_collect(tickets)          # (on the default stream)
        for k, s in zip(keeps, scenes):
            assert np.array_equal(k.cpu().numpy(), oracle.rnms(s, 0.5))
        del junk


def test_kept_rows_stay_differentiable_whatever_their_number():
    """dets[inds] of a large call went through the raw row-gather kernel and lost its autograd history."""
    from aidet_amd import rnms
    d = t(oracle.synth_boxes(20000, 4000, 9)).requires_grad_(True)
    kept, inds = rnms(d, 0.5)
    assert inds.numel() >= 4096 and kept.requires_grad
    kept[:, :2].sum().backward()
    assert float(d.grad[inds, 0].min()) == 1.0 and float(d.grad.sum()) == 2.0 * inds.numel()
    with torch.no_grad():
        k2, _ = rnms(d, 0.5)
    assert torch.equal(k2, kept.detach())


@pytest.mark.parametrize('kind,K', [(0, 512), (1, 2000), (2, 300), (0, 70000)])
def test_roi_forward_with_workspace_equals_the_plain_entry_points(kind, K):
    """obb_roi_align_forward_ws_f32 visits the RoIs in Morton order of their centres (rroi_order_kernel); which block
    computes a RoI must not change a bit of the output.  K = 70000 is beyond the order kernel's range (identity)."""
    from aidet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(K)
    C, H, W = (32, 96, 128) if K < 10000 else (8, 64, 64)
    feat = t(rng.standard_normal((2, C, H, W)).astype(np.float32))
    if kind == 0:
        r = oracle.synth_boxes(K, 4 * W, 3)[:, :5]
        r[:, 1] *= H / W
        rois = np.concatenate([rng.integers(0, 2, (K, 1)).astype(np.float32), r], 1)
    else:
        x1, y1 = rng.uniform(0, 4 * W - 40, K), rng.uniform(0, 4 * H - 40, K)
        rois = np.stack([rng.integers(0, 2, K), x1, y1, x1 + rng.uniform(4, 160, K), y1 + rng.uniform(4, 160, K)], 1).astype(np.float32)
    rt = t(rois)
    st = _lib.current_stream_handle(dev())
    plain = torch.empty((K, C, 7, 7), device=dev())
    if kind == 0:
        _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 2, C, H, W, rt.data_ptr(), K, 7, 7, 0.25, 2, 1.0, plain.data_ptr(), st))
    else:
        _lib.check(L.obb_roi_align_forward_f32(feat.data_ptr(), 2, C, H, W, rt.data_ptr(), K, 7, 7, 0.25, 2, 1 if kind == 2 else 0, plain.data_ptr(), st))
    ws = torch.empty(L.obb_roi_forward_workspace_bytes(K), dtype=torch.uint8, device=dev())
    ptr = (ctypes.c_void_p * 1)(feat.data_ptr())
    hh, ww, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
    for order_min in (64, 1 << 40):
        _lib.set_tuning(_lib.TUNE_RROI_ORDER_MIN, order_min)
        out = torch.full((K, C, 7, 7), float('nan'), device=dev())
        _lib.check(L.obb_roi_align_forward_ws_f32(ptr, hh, ww, sc, 1, 2, C, rt.data_ptr(), None, K, 7, 7, 2, kind, 1.0, out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), st))
        assert torch.equal(out, plain)
    _lib.set_tuning(_lib.TUNE_RROI_ORDER_MIN, -1)
    if K <= 65536:   # the order the kernel left in the workspace is a permutation of the RoIs
        order = ws[:4 * K].view(torch.int32)
        _lib.check(L.obb_roi_align_forward_ws_f32(ptr, hh, ww, sc, 1, 2, C, rt.data_ptr(), None, K, 7, 7, 2, kind, 1.0, out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), st))
        assert torch.equal(torch.sort(order.long())[0], torch.arange(K, device=dev()))
    # too small a workspace is refused, none at all is allowed
    assert L.obb_roi_align_forward_ws_f32(ptr, hh, ww, sc, 1, 2, C, rt.data_ptr(), None, K, 7, 7, 2, kind, 1.0, out.data_ptr(),
                                          ws.data_ptr(), 8, st) == _lib.OBB_EWORKSPACE
    _lib.check(L.obb_roi_align_forward_ws_f32(ptr, hh, ww, sc, 1, 2, C, rt.data_ptr(), None, K, 7, 7, 2, kind, 1.0, out.data_ptr(), None, 0, st))
    assert torch.equal(out, plain)
    if K <= 2000:    # ... and equal the oracle
        if kind == 0:
            ref = oracle.rroi_align_forward(feat.cpu().numpy(), rois, 7, 0.25, 2)
        else:
            ref = oracle.roi_align_forward(feat.cpu().numpy(), rois, 7, 0.25, 2, version=2 if kind == 2 else 1, aligned=kind == 2)
        assert np.array_equal(out.cpu().numpy(), ref) or np.abs(out.cpu().numpy() - ref).max() <= 1e-4


def test_grouped_soft_nms_equals_one_call_per_group(golden):
    """obb_soft_nms_grouped_f32: one workgroup per group, every group an array of its own = the per-class calls of
    multiclass_nms_with_index (rbbox_nms.py:29-49).  Index-exact (ties included: the scores are quantised)."""
    from aidet_amd import nms_wrapper, soft_nms
    rng = np.random.default_rng(3)
    sizes = [0, 37, 1, 300, 0, 768, 129, 5]
    groups = []
    for i, m in enumerate(sizes):
        x1, y1 = rng.uniform(0, 300, m), rng.uniform(0, 300, m)
        sc = np.round(rng.uniform(0, 1, m) * 50) / 50            # many equal scores
        groups.append(np.stack([x1, y1, x1 + rng.uniform(5, 80, m), y1 + rng.uniform(5, 80, m), sc], 1).astype(np.float32))
    allrows = np.concatenate(groups, 0)
    for method, kw in (('linear', dict(iou_thr=0.3, min_score=0.05)), ('gaussian', dict(iou_thr=0.3, sigma=0.4, min_score=0.02))):
        res = nms_wrapper.soft_nms_grouped(t(allrows), sizes, method=method, **kw)
        assert len(res) == len(sizes)
        for g, (d, k) in zip(groups, res):
            rd, ri = oracle.soft_nms(g, kw['iou_thr'], method, kw.get('sigma', 0.5), kw['min_score'])
            assert np.array_equal(k.cpu().numpy(), ri)
            assert d.shape == rd.shape
            assert np.array_equal(d.cpu().numpy(), rd)                # (gaussian too: glibc's expf restated on the device)
            if len(g):    # ... and the single-group entry point
                d1, k1 = soft_nms(t(g), kw['iou_thr'], method=method, sigma=kw.get('sigma', 0.5), min_score=kw['min_score'])
                assert torch.equal(k1, k) and torch.equal(d1, d)
    # groups beyond the LDS variant (n > 2048 in all): state in the workspace
    big = [groups[5]] * 4
    res = nms_wrapper.soft_nms_grouped(t(np.concatenate(big, 0)), [len(b) for b in big], 0.3, min_score=0.05)
    rd, ri = oracle.soft_nms(groups[5], 0.3, 'linear', 0.5, 0.05)
    for d, k in res:
        assert np.array_equal(k.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd)
    assert nms_wrapper.soft_nms_grouped(t(np.zeros((0, 5), np.float32)), [], 0.3) == []


def test_multiclass_soft_nms_is_one_launch_and_still_reproduces_the_reference(golden):
    """multiclass_nms_with_index(type='soft_nms') on the reference's own fixture, through the grouped launch; and the
    grouped branch of multiclass_nms (it raised a TypeError for soft_nms)."""
    from aidet_amd import multiclass_nms, multiclass_nms_with_index
    g, s = golden('multiclass.npz'), golden('multiclass_soft.npz')
    mb, ms = t(g['multi_bboxes']), t(g['multi_scores'])
    cfg = dict(type='soft_nms', iou_thr=0.5, min_score=0.05)
    boxes, labels, cls_inds, keep_inds = multiclass_nms_with_index(mb, ms, 0.05, cfg, 100)
    assert np.array_equal(labels.cpu().numpy(), s['idx_labels']) and np.allclose(boxes.cpu().numpy(), s['idx_bboxes'], atol=1e-6)
    for i, k in enumerate(keep_inds):
        assert np.array_equal(k.cpu().numpy(), s['idx_keep_%d' % i])
    b2, l2 = multiclass_nms(mb, ms, 0.05, cfg, 100)                       # the reference's offset trick: one global array
    assert np.array_equal(l2.cpu().numpy(), s['mc_labels']) and np.allclose(b2.cpu().numpy(), s['mc_bboxes'], atol=1e-6)
    b3, l3 = multiclass_nms(mb, ms, 0.05, cfg, 100, grouped=True)         # every class an array of its own
    assert b3.shape == b2.shape and sorted(l3.tolist()) == sorted(l2.tolist())
    assert np.allclose(np.sort(b3[:, 4].cpu().numpy()), np.sort(b2[:, 4].cpu().numpy()), atol=1e-6)


def test_per_box_thresholds_of_horizontal_nms_must_be_positive():
    """a per-box threshold <= 0 under `>=` is the reference's suppress-everything rule, not served."""
    from aidet_amd import nms
    from aidet_amd import dota_merge
    h = np.array([[0, 0, 10, 10, .9], [20, 20, 30, 30, .8], [0, 0, 10, 10, .7]], np.float32)
    with pytest.raises(ValueError):
        nms(t(h), 0.5, thrs=np.array([0.5, 0.0, 0.5], np.float32))
    kept, inds = nms(t(h), 0.5, thrs=np.array([0.5, 0.5, 0.5], np.float32))
    assert inds.tolist() == [0, 1]
    rows = np.concatenate([h, np.zeros((3, 2), np.float32)], 1)            # 7 columns: horizontal merge rows
    with pytest.raises(ValueError):
        dota_merge.merge_detections(t(rows), 0.3, 2, class_thrs=[0.0, 0.3])


@pytest.mark.parametrize('case', ['k2049', 'k5000', 'tiny', 'edge2047', 'edge2048'])
def test_gather_backward_list_batches_long_lists_and_capacity_edges(case):
    """the gather backward beyond its first unit test -- more RoIs than one LDS list batch (2048), pixels with
    many taps (tiny RoIs: the plan's long-list queue and bit-map ranks), and footprints of exactly 2047 (planned) and
    2048 (left to the atomics kernel) pixels.  Gather vs atomics vs oracle; the gather is bit-reproducible."""
    from aidet_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7)
    C, H, W, scale, sn = 8, 64, 96, 1.0, 2
    if case in ('k2049', 'k5000'):
        K = 2049 if case == 'k2049' else 5000
        r = oracle.synth_boxes(K, 90, 5)[:, :5]
        r[:, 2:4] = np.clip(r[:, 2:4] * 0.25, 2, 30)
        r[:, 1] *= H / 90.0
    elif case == 'tiny':
        K = 300
        r = oracle.synth_boxes(K, 90, 6)[:, :5]
        r[:, 2:4] = rng.uniform(0.05, 2.0, (K, 2))          # sub-pixel to 2 x 2 px: up to 196 taps on one pixel
        r[:, 1] *= H / 90.0
    else:
        # an upright RoI whose 2 x 2 x 49 samples span exactly 23 x 89 = 2047 (or 32 x 64 = 2048) pixels
        K = 1
        # (the 14 sample positions per axis start w / 28 inside the RoI and are w / 14 apart: first sample at 2.25, last at
        # fw + 0.5 -> the taps span the pixels 2 .. fw + 1, i.e. fw of them)
        fw, fh = (89, 23) if case == 'edge2047' else (64, 32)
        w, h = (fw - 1.75) * 14 / 13.0, (fh - 1.75) * 14 / 13.0
        r = np.array([[2.25 + w * 13 / 28.0, 2.25 + h * 13 / 28.0, w, h, 0.0]], np.float32)
    rois = np.concatenate([np.zeros((K, 1), np.float32), r.astype(np.float32)], 1)
    g = rng.standard_normal((K, C, 7, 7)).astype(np.float32)
    gt, rt = t(g), t(rois)
    st = _lib.current_stream_handle(dev())
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev())
    outs = []
    for rep in range(2):
        gm = torch.full((1, C, H, W), float('nan'), device=dev())
        _lib.check(L.obb_roi_backward_gather_f32(gt.data_ptr(), 1, C, H, W, rt.data_ptr(), K, 0, 7, 7, scale, sn, 1.0, gm.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), st))
        outs.append(gm.clone())
    assert torch.equal(outs[0], outs[1])                                   # bit-reproducible
    at = torch.zeros((1, C, H, W), device=dev())
    _lib.check(L.obb_rroi_align_backward_f32(gt.data_ptr(), 1, C, H, W, rt.data_ptr(), K, 7, 7, scale, sn, 1.0, at.data_ptr(), st))
    ref = oracle.rroi_align_backward(g, rois, (1, C, H, W), scale, sn)
    tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[0].cpu().numpy() - ref).max() <= tol
    assert np.abs(at.cpu().numpy() - ref).max() <= tol
    if case.startswith('edge'):     # the taps' bounding box really is fw x fh pixels
        ys, xs = np.nonzero(ref[0, 0])
        assert (xs.max() - xs.min() + 1, ys.max() - ys.min() + 1) == (fw, fh)


@pytest.mark.parametrize('sigmoid', [True, False])
def test_fused_rpn_selection_equals_the_tensor_op_path(sigmoid):
    """obb_rpn_select_decode_f32 + grouped nms + obb_rows_topk_f32 against the tensor-op path of rpn.py (the
    restatement of rpn_head.py:55-104 that detector_parts.npz pins): the same proposals in the same order on a 1024^2
    tile's five levels, also with an nms_post cut and a max_num below the survivors."""
    from aidet_amd import rpn
    from aidet_amd.anchors import AnchorGenerator
    g = torch.Generator().manual_seed(5)
    strides = [4, 8, 16, 32, 64]
    A = 3
    cls = [torch.randn((A if sigmoid else 2 * A, 256 // (s // 4), 256 // (s // 4)), generator=g).to(dev()) for s in strides]
    reg = [(0.3 * torch.randn((4 * A, c.size(1), c.size(2)), generator=g)).to(dev()) for c in cls]
    anchors = [AnchorGenerator(s, [8], [0.5, 1.0, 2.0]).grid_anchors(c.shape[-2:], s, device=dev()) for s, c in zip(strides, cls)]
    for cfg in (dict(nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7, min_bbox_size=0, nms_across_levels=False),
                dict(nms_pre=1000, nms_post=300, max_num=700, nms_thr=0.6, min_bbox_size=0, nms_across_levels=False)):
        rpn.FUSED = True
        got = rpn.rpn_proposals(cls, reg, anchors, (1024, 1024, 3), cfg, use_sigmoid_cls=sigmoid)
        rpn.FUSED = False
        try:
            want = rpn.rpn_proposals(cls, reg, anchors, (1024, 1024, 3), cfg, use_sigmoid_cls=sigmoid)
        finally:
            rpn.FUSED = True
        assert got.shape == want.shape and got.shape[0] == min(cfg['max_num'], want.shape[0])
        assert bool((got[:-1, 4] >= got[1:, 4]).all()) and torch.allclose(got[:, 4], want[:, 4], atol=1e-6)
        # the kernel orders by the LOGIT (l1 - l0 for softmax scores), torch by the ROUNDED scores: anchors whose scores
        # agree to the last bits may trade places (torch.topk leaves the order of equal scores open anyway); the
        # proposals are the same set -- up to the members of a tie that straddles the cut
        assert same_rows_up_to_score_ties(got, want)
    # the selection kernel alone against torch.topk on one level, incl. equal scores (ties: ascending anchor index)
    from aidet_amd import _lib
    L = _lib.lib()
    c1 = torch.round(torch.randn((A, 64, 64), generator=g) * 4).to(dev()) / 4         # many equal logits
    r1 = torch.zeros((4 * A, 64, 64), device=dev())
    a1 = AnchorGenerator(8, [8], [0.5, 1.0, 2.0]).grid_anchors((64, 64), 8, device=dev())
    dets = torch.empty((500, 5), device=dev()); lv = torch.empty(500, dtype=torch.int32, device=dev())
    rows = ctypes.c_int64(0)
    _lib.check(L.obb_rpn_select_decode_f32((ctypes.c_void_p * 1)(c1.data_ptr()), (ctypes.c_void_p * 1)(r1.data_ptr()),
                                           (ctypes.c_void_p * 1)(a1.data_ptr()), (ctypes.c_int * 1)(64), (ctypes.c_int * 1)(64), 1, A, 1, 500,
                                           (ctypes.c_float * 4)(0, 0, 0, 0), (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 0, 0,
                                           dets.data_ptr(), lv.data_ptr(), ctypes.byref(rows), _lib.current_stream_handle(dev())))
    flat = c1.permute(1, 2, 0).reshape(-1)
    order = torch.sort(flat, descending=True, stable=True)[1][:500]       # stable: equal logits by ascending index
    assert rows.value == 500 and torch.allclose(dets[:, 4], flat[order].sigmoid(), atol=1e-6)
    assert torch.allclose(dets[:, :4], a1[order], atol=1e-4)              # zero deltas: the anchors themselves


def test_hip_nms_and_soft_nms_against_the_reference_build_live():
    """oracle/_ref/nms_cpu.so = the reference's own mmdet/ops/nms/src/nms_cpu.cpp, compiled in the build container by
    oracle/build_ref.py; it travels with the snapshot (not its sources).  Where it is present the HIP `nms` / `soft_nms`
    are compared with it LIVE on fresh boxes -- beyond the committed fixtures: float boxes, integer boxes (IoU == thr
    happens), 20 000 boxes through the sorted route."""
    from oracle import build_ref
    ref = build_ref.load_ref()
    if ref is None:
        pytest.skip('oracle/_ref/nms_cpu.so is not part of this snapshot')
    from aidet_amd import nms, soft_nms
    rng = np.random.default_rng(2026)
    for n, integer, thr in ((1500, False, 0.5), (3000, True, 0.5), (20000, False, 0.7), (20000, True, 0.3)):
        x1, y1 = rng.uniform(0, 1500, n), rng.uniform(0, 1500, n)
        b = np.stack([x1, y1, x1 + rng.uniform(4, 90, n), y1 + rng.uniform(4, 90, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        if integer:
            b[:, :4] = np.round(b[:, :4] / 4) * 4
            b[:, 4] = np.round(b[:, 4] * 200) / 200       # tied scores too: the reference sorts unstably, so compare sets of survivors
        want = ref.nms(torch.from_numpy(b), float(thr)).numpy()
        got = nms(t(b), thr)[1].cpu().numpy()
        if integer:
            # equal scores: which of two tied, mutually suppressing boxes survives depends on the reference's unstable sort
            assert abs(len(got) - len(want)) <= max(3, len(want) // 200)
        else:
            assert np.array_equal(got, want)
    x1, y1 = rng.uniform(0, 300, 800), rng.uniform(0, 300, 800)
    b = np.stack([x1, y1, x1 + rng.uniform(5, 80, 800), y1 + rng.uniform(5, 80, 800), rng.uniform(0, 1, 800)], 1).astype(np.float32)
    for method in (1, 2):
        res = ref.soft_nms(torch.from_numpy(b), 0.3, method, 0.5, 0.05).numpy()      # (k, 6): box, decayed score, original index
        gd, gi = soft_nms(t(b), 0.3, method={1: 'linear', 2: 'gaussian'}[method], sigma=0.5, min_score=0.05)
        assert np.array_equal(gi.cpu().numpy(), res[:, 5].astype(np.int64))
        assert np.array_equal(gd.cpu().numpy(), res[:, :5])           # (bit-equal to the compiled reference, gaussian weights included)


# ---- row routing / order restoration kernels (config 5 over several ranks) ------------------------------------------
def _route_spec(rows, world, key):
    """the tensor-op specification (dota_merge.route_rows on CPU tensors)"""
    from aidet_amd import dota_merge
    out, counts = dota_merge.route_rows(rows.cpu(), world, None if key is None else key.cpu())
    return out, counts


@pytest.mark.gpu
@pytest.mark.parametrize('n,world,cols', [(0, 4, 8), (1, 1, 8), (1000, 2, 8), (1024, 8, 8), (5000, 3, 11), (70000, 8, 8), (4097, 64, 7)])
def test_route_rows_matches_the_tensor_ops(n, world, cols):
    from aidet_amd import dota_merge
    g = torch.Generator().manual_seed(n + world)
    rows = torch.randn(n, cols, generator=g)
    rows[:, -1] = torch.randint(-2, 40, (n,), generator=g).float()      # image ids (negative: rows of no image)
    key = torch.sort(torch.randint(0, 300, (n,), generator=g))[0]
    for k in (key, None):
        out, counts = dota_merge.route_rows(rows.cuda(), world, None if k is None else k.cuda())
        ref, rcounts = _route_spec(rows, world, k)
        assert torch.equal(counts.cpu(), rcounts)
        assert torch.equal(out.cpu(), ref)
    if n > 10:      # an unordered key is counted, not repaired
        bad = key.clone(); bad[5] = 400
        _, counts = dota_merge.route_rows(rows.cuda(), world, bad.cuda())
        assert int(counts[-1]) == int((bad[1:] < bad[:-1]).sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('world,per,kmax', [(1, 1000, 50), (2, 3000, 200), (8, 20000, 4000), (3, 0, 10), (64, 300, 2000), (5, 700, 32767)])
def test_restore_rows_is_a_stable_merge_of_the_sources(world, per, kmax):
    from aidet_amd import dota_merge
    g = torch.Generator().manual_seed(world * 7 + per)
    segs, starts = [], []
    nmax = per + 5
    for r in range(world):
        m = int(torch.randint(0, per + 1, (1,), generator=g)) if per else 0
        if r == 1:
            m = 0                                              # an empty source
        body = torch.randn(m, 9, generator=g)
        # a source's keys ascend; two sources may hold the same key (a stable merge orders them by source)
        body[:, -1] = torch.sort(torch.randint(0, kmax + 1, (m,), generator=g))[0].float()
        pad = torch.zeros(nmax - m, 9); pad[:, -1] = dota_merge._PAD_KEY
        segs.append(torch.cat([body, pad]))
        starts.append(r * nmax)
    rows = torch.cat(segs)
    total = int((rows[:, -1] <= kmax).sum())
    ref = rows[torch.argsort(rows[:, -1].to(torch.int32), stable=True)[:total]]
    for oc in (9, 8):
        out = dota_merge.restore_rows(rows.cuda(), starts, kmax, oc, total)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref[:, :oc])
    dota_merge._check_order_flag(torch.device('cuda', 0))       # nothing to report
    assert torch.equal(dota_merge._restore_by_key(rows.cuda(), total, 8, kmax, world).cpu(), ref[:, :8])


@pytest.mark.gpu
def test_restore_rows_reports_a_source_that_is_out_of_order():
    from aidet_amd import dota_merge
    rows = torch.zeros(100, 9)
    rows[:, -1] = torch.arange(100).float()
    rows[40, -1] = 7.0                                          # smaller than its predecessor's 39
    dota_merge.restore_rows(rows.cuda(), [0, 50], 200, 9, 100)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match='non-decreasing'):
        dota_merge._check_order_flag(torch.device('cuda', 0))
    dota_merge._check_order_flag(torch.device('cuda', 0))       # (the flag is cleared by the report)


@pytest.mark.gpu
def test_route_and_restore_refuse_bad_arguments():
    from aidet_amd import _lib
    L = _lib.lib()
    r = torch.zeros(10, 8, device='cuda')
    c = torch.zeros(70, dtype=torch.int64, device='cuda')
    o = torch.zeros(10, 9, device='cuda')
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device='cuda')
    assert L.obb_rows_route_f32(r.data_ptr(), 10, 8, 7, 65, None, o.data_ptr(), c.data_ptr(), ws.data_ptr(), ws.numel(), None) == _lib.OBB_EINVAL
    assert L.obb_rows_route_f32(r.data_ptr(), 10, 8, 8, 4, None, o.data_ptr(), c.data_ptr(), ws.data_ptr(), ws.numel(), None) == _lib.OBB_EINVAL
    assert L.obb_rows_route_f32(r.data_ptr(), 10, 8, 7, 4, None, o.data_ptr(), c.data_ptr(), ws.data_ptr(), 4, None) == _lib.OBB_EWORKSPACE
    seg = (ctypes.c_int64 * 2)(0, 20)
    assert L.obb_rows_restore_f32(r.data_ptr(), 10, 8, 7, seg, 2, 100, 8, 10, o.data_ptr(), None, ws.data_ptr(), ws.numel(), None) == _lib.OBB_EINVAL
    assert L.obb_rows_restore_workspace_bytes(64, 1 << 20) == 0


@pytest.mark.parametrize('case', ['random', 'ties', 'constant', 'softmax', 'ragged'])
def test_rpn_selection_over_several_workgroups_is_exact(case):
    """obb_rpn_select_decode_ws_f32: levels of two and more 8192-anchor slices are selected by all their workgroups
    together (rpn_group_select_kernel: one read of the level, a barrier per radix digit).  The result is the stable
    descending sort's first nms_pre anchors -- equal logits by ascending anchor index, also when thousands of them sit
    AT the cut -- and equals the one-workgroup-per-level path wherever that one is defined the same way."""
    from aidet_amd import _lib
    from aidet_amd.anchors import AnchorGenerator
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    A = 3
    sizes = [(256, 256), (128, 128), (64, 64), (32, 32)] if case != 'ragged' else [(200, 333), (53, 61), (47, 59)]
    sig = case != 'softmax'
    cls = []
    for (h, w) in sizes:
        c = torch.randn((A if sig else 2 * A, h, w), generator=g)
        if case == 'ties':
            c = torch.round(c * 2) / 2                      # ~20 distinct values: the cut falls inside a huge tie
        if case == 'constant':
            c = torch.zeros_like(c)
        cls.append(c.to(dev()))
    reg = [torch.zeros((4 * A, c.size(1), c.size(2)), device=dev()) for c in cls]
    anchors = [AnchorGenerator(4 << i, [8], [0.5, 1.0, 2.0]).grid_anchors(c.shape[-2:], 4 << i, device=dev()) for i, c in enumerate(cls)]
    n = len(cls)
    hs = (ctypes.c_int * n)(*[c.size(1) for c in cls]); wd = (ctypes.c_int * n)(*[c.size(2) for c in cls])
    for nms_pre in (2000, 777):
        per = [min(nms_pre, c.size(1) * c.size(2) * A) for c in cls]
        N = sum(per)
        wsb = int(L.obb_rpn_select_workspace_bytes(hs, wd, n, A, nms_pre))
        assert wsb > 0
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
        outs = []
        for use_ws in (True, False):
            dets = torch.empty((N, 5), device=dev()); lv = torch.empty(N, dtype=torch.int32, device=dev())
            rows = ctypes.c_int64(0)
            _lib.check(L.obb_rpn_select_decode_ws_f32((ctypes.c_void_p * n)(*[c.data_ptr() for c in cls]), (ctypes.c_void_p * n)(*[r.data_ptr() for r in reg]),
                                                      (ctypes.c_void_p * n)(*[a.data_ptr() for a in anchors]), hs, wd, n, A, 1 if sig else 0, nms_pre,
                                                      (ctypes.c_float * 4)(0, 0, 0, 0), (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 0, 0,
                                                      dets.data_ptr(), lv.data_ptr(), ctypes.byref(rows),
                                                      ws.data_ptr() if use_ws else None, wsb if use_ws else 0, _lib.current_stream_handle(dev())))
            assert rows.value == N
            outs.append((dets, lv))
        got, glv = outs[0]
        base = 0
        for l, c in enumerate(cls):
            flat = c.permute(1, 2, 0).reshape(-1) if sig else (c[1::2] - c[0::2]).permute(1, 2, 0).reshape(-1)
            order = torch.sort(flat, descending=True, stable=True)[1][:per[l]] if per[l] < flat.numel() else torch.arange(flat.numel(), device=dev())
            assert bool((glv[base:base + per[l]] == l).all())
            assert torch.allclose(got[base:base + per[l], 4], flat[order].sigmoid(), atol=1e-6), (case, l)
            assert torch.equal(got[base:base + per[l], :4], anchors[l][order]), (case, l)     # zero deltas, no clamp: the anchors
            base += per[l]
        if case in ('random', 'softmax', 'ragged'):      # (no ties at the cut: both paths are defined alike)
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the second call reuses the workspace as it was left (the entry point zeroes what it needs)


def test_level_mapping_kernel_equals_the_tensor_expressions():
    """obb_roi_map_levels_i32 against SingleRoIExtractor.map_roi_levels / RotatedSingleRoIExtractor.map_roi_levels
    (single_level.py:54-73), incl. scales exactly on a level boundary (56 * 2^k), tiny and huge RoIs."""
    from aidet_amd import roi_layers
    g = torch.Generator().manual_seed(3)
    K = 5000
    x1, y1 = torch.rand(K, generator=g) * 900, torch.rand(K, generator=g) * 900
    side = torch.exp(torch.rand(K, generator=g) * 7.0)            # 1 .. 1100 px
    side[:8] = torch.tensor([56.0, 112.0, 224.0, 448.0, 55.999, 112.001, 0.5, 5000.0])
    hb = torch.stack([torch.zeros(K), x1, y1, x1 + side - 1, y1 + side * (0.5 + torch.rand(K, generator=g)) - 1], 1).to(dev())
    rb = torch.stack([torch.zeros(K), x1, y1, side, side * (0.5 + torch.rand(K, generator=g)), torch.zeros(K)], 1).to(dev())
    for n in (4, 1):
        ext = roi_layers.SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=2), 8, [4, 8, 16, 32][:n])
        assert torch.equal(roi_layers.map_levels(hb, n, ext.finest_scale, 1).long(), ext.map_roi_levels(hb, n))
        rext = roi_layers.RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=7, sample_num=2), 8, [4, 8, 16, 32][:n])
        assert torch.equal(roi_layers.map_levels(rb, n, rext.finest_scale, 0).long(), rext.map_roi_levels(rb, n))
    assert roi_layers.map_levels(hb[:0], 4, 56, 1).shape == (0,)


def test_grouped_soft_nms_keeps_lds_state_for_many_small_groups():
    """obb_soft_nms_grouped_max_f32: 15 groups of ~400 rows (6000 in all, beyond the 2048 rows one LDS state holds) run
    on the LDS variant because every GROUP fits, with the results of one call per group; a bound above the row count is
    clamped to it."""
    from aidet_amd import _lib
    from aidet_amd.nms_wrapper import soft_nms, soft_nms_grouped
    rng = np.random.default_rng(9)
    parts, sizes = [], []
    for c in range(15):
        m = int(rng.integers(300, 420))
        x1, y1 = rng.uniform(0, 600, m), rng.uniform(0, 600, m)
        parts.append(np.stack([x1, y1, x1 + rng.uniform(8, 150, m), y1 + rng.uniform(8, 150, m), rng.uniform(0.05, 1, m)], 1).astype(np.float32))
        sizes.append(m)
    d = t(np.concatenate(parts))
    got = soft_nms_grouped(d, sizes, 0.5, min_score=0.05)
    o = 0
    for m, (gd, gi) in zip(sizes, got):
        wd, wi = soft_nms(d[o:o + m], 0.5, min_score=0.05)
        assert torch.equal(gi, wi) and torch.equal(gd, wd)
        o += m
    # a bound above n
    L = _lib.lib()
    n = d.size(0)
    seg = torch.tensor([0, 100, 1000], dtype=torch.int64, device=dev())
    out = torch.full((n, 5), -7.0, device=dev()); oi = torch.zeros(n, dtype=torch.int64, device=dev())
    counts = torch.zeros(2, dtype=torch.int64, device=dev())
    ws = torch.empty(int(L.obb_soft_nms_workspace_bytes(n)), dtype=torch.uint8, device=dev())
    _lib.check(L.obb_soft_nms_grouped_max_f32(d.data_ptr(), 1000, seg.data_ptr(), 2, 4000, 0.5, 1, 0.5, 0.05, out.data_ptr(), oi.data_ptr(),
                                              counts.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_handle(dev())))
    torch.cuda.synchronize()
    assert int(counts[0]) > 0 and int(counts[1]) > 0          # (a bound above n is clamped to n: both groups run)
