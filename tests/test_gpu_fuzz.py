"""Bounded randomised parity sweeps on the GPU (fixed seeds; part of `pytest -m gpu`, so the
driver runs them): >= 300 NMS scenes x 5 routes and >= 300 IoU / RoI-Align configurations
against the oracle.  (Formerly one-off scripts under tools/.)  Sized to finish in about a
minute; the oracle's O(n^2) loop is what bounds n."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nms_scene(it, n_max=3000):
    """clusters, rounded (tied) scores, duplicates, far outliers, arbitrary group ids, per-box
    thresholds -- the generator of the former tools/fuzz_nms.py"""
    rng = np.random.default_rng(1000 + it)
    n = int(np.exp(rng.uniform(np.log(2), np.log(n_max))))
    k = int(rng.integers(1, 30))
    centres = rng.uniform(0, rng.choice([50, 500, 5000]), (k, 2)) + rng.choice([0.0, 1e4, -3e5])
    which = rng.integers(0, k, n)
    d = np.zeros((n, 6), np.float32)
    d[:, :2] = centres[which] + rng.normal(0, rng.uniform(1, 100), (n, 2))
    d[:, 2] = np.exp(rng.uniform(np.log(1), np.log(rng.choice([30, 300])), n))
    d[:, 3] = np.exp(rng.uniform(np.log(1), np.log(rng.choice([30, 100])), n))
    d[:, 4] = rng.uniform(-np.pi, np.pi, n) if it % 3 else rng.choice([0.0, -np.pi / 2, np.pi / 4], n)
    d[:, 5] = rng.random(n)
    if it % 2:
        d[:, 5] = np.round(d[:, 5], int(rng.integers(1, 3)))
    if it % 5 == 0 and n > 10:   # duplicates and outliers
        d[n // 2:n // 2 + n // 10] = d[:n // 10]
        d[rng.integers(0, n), 0] += 1e7
    thr = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95]))
    groups = thrs = None
    if it % 4 == 0:
        ng = int(rng.choice([2, 3, 15, 400, 5000]))
        ids = rng.integers(-2**31, 2**31 - 1, ng, dtype=np.int64).astype(np.int32)
        groups = ids[rng.integers(0, ng, n)]
    if it % 6 == 0:
        thrs = rng.choice([0.1, 0.3, 0.5, 0.7], n).astype(np.float32)
    return d, thr, groups, thrs


@pytest.mark.parametrize('block', range(6))
def test_fuzz_rnms_all_routes(block, nms_route):
    from aidet_amd import nms, poly_nms, rnms
    scenes = mismatches = 0
    for it in range(block * 52, block * 52 + 52):
        d, thr, groups, thrs = nms_scene(it)
        strict = it % 9 == 0
        ref = oracle.rnms(d, thr, groups=groups, thrs=thrs, strict=strict)
        dt = t(d)
        for route in ('auto', 'sorted', 'listed', 'direct', 'fused', 'grid'):
            if route == 'fused' and len(d) > 1024:
                continue
            nms_route(route)
            got = rnms(dt, thr, groups=groups, thrs=thrs, strict=strict)[1].cpu().numpy()
            mismatches += not np.array_equal(got, ref)
        nms_route('auto')
        if it % 7 == 0:
            p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
            mismatches += not np.array_equal(poly_nms(t(p), thr)[1].cpu().numpy(), oracle.poly_nms(p, thr))
        if it % 3 == 0:   # the same scene as horizontal boxes: the reference's exact arithmetic
            h = np.stack([d[:, 0] - d[:, 2], d[:, 1] - d[:, 3], d[:, 0] + d[:, 2], d[:, 1] + d[:, 3], d[:, 5]], 1)
            if it % 2:
                h[:, :4] = np.round(h[:, :4])        # integer boxes: IoU == thr happens
            h = h.astype(np.float32)
            for route in ('auto', 'sorted', 'listed', 'direct', 'grid'):
                nms_route(route)
                got = nms(t(h), thr, groups=groups, strict=strict)[1].cpu().numpy()
                mismatches += not np.array_equal(got, oracle.hbb_nms(h, thr, groups=groups, strict=strict))
            nms_route('auto')
        scenes += 1
    assert scenes == 52 and mismatches == 0


@pytest.mark.parametrize('block', range(6))
def test_fuzz_overlaps_and_rroi_align(block):
    from aidet_amd import rbbox_overlaps, rroi_align
    from aidet_amd.rbbox_geometry import pointobb_overlaps
    bad = []
    for it in range(block * 52, block * 52 + 52):
        rng = np.random.default_rng(5000 + it)
        m = int(np.exp(rng.uniform(0, np.log(1200))))
        n = int(np.exp(rng.uniform(0, np.log(2000))))
        canvas = float(rng.choice([40, 400, 4000]))
        a = oracle.synth_boxes(m, canvas, 10 * it + 1)[:, :5]
        b = oracle.synth_boxes(n, canvas, 10 * it + 2)[:, :5]
        off = rng.choice([0.0, 1e5])
        a[:, :2] += off
        b[:, :2] += off
        if rng.random() < 0.3:
            b[: min(m, n)] = a[: min(m, n)]             # identical boxes -> IoU 1
        if rng.random() < 0.2:
            a[rng.integers(0, m), 2] = 0.0              # degenerate
        mode = 'iou' if it % 2 else 'iof'
        got = rbbox_overlaps(t(a), t(b), mode=mode).cpu().numpy()
        ref = oracle.overlaps(a, b, mode)
        if not np.array_equal(got, ref):
            bad.append(('overlaps', it, float(np.abs(got - ref).max())))
        k = min(m, n)
        ga = rbbox_overlaps(t(a[:k]), t(b[:k]), mode=mode, is_aligned=True).cpu().numpy()
        if not np.array_equal(ga, np.diagonal(ref[:k, :k])):
            bad.append(('aligned', it))
        if it % 4 == 0:
            pa, pb = oracle.rbox_to_poly(a).astype(np.float32), oracle.rbox_to_poly(b).astype(np.float32)
            gp = pointobb_overlaps(t(pa), t(pb), mode=mode).cpu().numpy()
            if not np.array_equal(gp, oracle.overlaps(pa, pb, mode, kind='poly')):
                bad.append(('poly', it))
        # ---- rotated RoI-Align forward + backward
        B = int(rng.integers(1, 3)); C = int(rng.choice([1, 3, 8, 24, 40]))
        H = int(rng.integers(8, 90)); W = int(rng.integers(8, 90))
        K = int(rng.integers(1, 40)); ph = int(rng.integers(1, 9)); pw = int(rng.integers(1, 9))
        sn = int(rng.choice([0, 1, 2, 3]))
        scale = float(rng.choice([1.0, 0.5, 0.25, 0.125]))
        feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
        r = oracle.synth_boxes(K, max(W, H) / scale, 10 * it + 3)[:, :5]
        r[:, 2:4] *= rng.choice([0.2, 1.0, 3.0])
        rois = np.concatenate([rng.integers(0, B, (K, 1)).astype(np.float32), r], 1).astype(np.float32)
        ft = t(feat).requires_grad_(True)
        out = rroi_align(ft, t(rois), (ph, pw), scale, sn)
        ref = oracle.rroi_align_forward(feat, rois, (ph, pw), scale, sn)
        if np.abs(out.detach().cpu().numpy() - ref).max() > 1e-4 * max(1.0, np.abs(ref).max()):
            bad.append(('rroi fwd', it, float(np.abs(out.detach().cpu().numpy() - ref).max())))
        g = rng.standard_normal(out.shape).astype(np.float32)
        out.backward(t(g))
        rb = oracle.rroi_align_backward(g, rois, feat.shape, scale, sn)
        if np.abs(ft.grad.cpu().numpy() - rb).max() > 1e-4 * max(1.0, np.abs(rb).max()):
            bad.append(('rroi bwd', it, float(np.abs(ft.grad.cpu().numpy() - rb).max())))
    assert not bad, bad


@pytest.mark.parametrize('block', range(4))
def test_fuzz_channels_last_roi_align(block):
    """Round 5: random shapes through the channels-last kernels -- forward bit-equal to the NCHW kernel and <= 1e-4 from
    the oracle; backward <= 1e-4 max|g| from the oracle and (every RoI planned, fixed sampling grid) bit-equal to the NCHW
    gather; rotated and horizontal kinds, C any multiple of 4, tiles overhanging the map, several images."""
    from aidet_amd import RRoIAlign, RoIAlign, roi_layers
    bad = []
    for it in range(block * 40, block * 40 + 40):
        rng = np.random.default_rng(9000 + it)
        B = int(rng.integers(1, 4)); C = 4 * int(rng.integers(1, 80))
        H = int(rng.integers(3, 70)); W = int(rng.integers(3, 70))
        K = int(rng.integers(1, 60)); ph = int(rng.integers(1, 9)); pw = int(rng.integers(1, 9))
        sn = int(rng.choice([0, 1, 2, 2, 3]))
        scale = float(rng.choice([1.0, 0.5, 0.25]))
        feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
        r = oracle.synth_boxes(K, max(W, H) / scale, 10 * it + 3)[:, :5]
        r[:, 2:4] *= rng.choice([0.1, 1.0, 2.5])
        rois = np.concatenate([rng.integers(0, B, (K, 1)).astype(np.float32), r], 1).astype(np.float32)
        layer = RRoIAlign((ph, pw), scale, sample_num=sn)
        fa = t(feat).requires_grad_(True)
        fb = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        if not roi_layers.is_nhwc(fb):        # (one pixel or one channel: both layouts are the same memory)
            continue
        oa, ob = layer(fa, t(rois)), layer(fb, t(rois))
        if not torch.equal(oa, ob):
            bad.append(('fwd vs nchw', it, C, H, W, K, ph, pw, sn))
        ref = oracle.rroi_align_forward(feat, rois, (ph, pw), scale, sn)
        if np.abs(ob.detach().cpu().numpy() - ref).max() > 1e-4 * max(1.0, np.abs(ref).max()):
            bad.append(('fwd vs oracle', it))
        g = rng.standard_normal(ob.shape).astype(np.float32)
        oa.backward(t(g)); ob.backward(t(g))
        rb = oracle.rroi_align_backward(g, rois, feat.shape, scale, sn)
        if np.abs(fb.grad.cpu().numpy() - rb).max() > 1e-4 * max(1.0, np.abs(rb).max()):
            bad.append(('bwd vs oracle', it, float(np.abs(fb.grad.cpu().numpy() - rb).max())))
        if sn > 0 and ph * pw <= 64 and ph * pw * sn * sn <= 256 and K >= roi_layers.GATHER_MIN_ROIS and not torch.equal(fb.grad.contiguous(), fa.grad):
            small = (r[:, 2] * scale * 1.5 + 3) * (r[:, 3] * scale * 1.5 + 3) < 2000     # (RoIs the plan holds on both sides)
            if small.all():
                bad.append(('bwd vs nchw gather', it, C, H, W, K))
        if it % 4 == 0:      # horizontal kinds on the same map
            x1 = rng.uniform(0, W / scale, K); y1 = rng.uniform(0, H / scale, K)
            hr = np.stack([rois[:, 0], x1, y1, x1 + rng.uniform(1, W / scale, K), y1 + rng.uniform(1, H / scale, K)], 1).astype(np.float32)
            for aligned in (False, True):
                hl = RoIAlign((ph, pw), scale, sample_num=sn, aligned=aligned)
                a, b = hl(t(feat), t(hr)), hl(fb.detach(), t(hr))
                if not torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0)):
                    bad.append(('horizontal fwd', it, aligned))
    assert not bad, bad


def test_fuzz_nms_nonpositive_thresholds():
    """`nms(dets, thr <= 0)` against the oracle's restatement of nms_cpu.cpp:5-68 on random horizontal scenes: float and
    double rows, inverted and zero-area boxes, groups, thresholds 0, -0.5, NaN."""
    from aidet_amd import nms
    bad = []
    for it in range(60):
        rng = np.random.default_rng(7000 + it)
        n = int(np.exp(rng.uniform(0, np.log(3000))))
        x1, y1 = rng.uniform(0, 300, n), rng.uniform(0, 300, n)
        h = np.stack([x1, y1, x1 + rng.uniform(-5, 60, n), y1 + rng.uniform(-5, 60, n), (rng.permutation(n) + 0.5) / n], 1)
        if it % 3 == 0:
            h[:, :4] = np.round(h[:, :4])
            h[rng.integers(0, n, max(n // 10, 1)), 2] = h[rng.integers(0, n, max(n // 10, 1)), 0] - 1      # zero-area boxes
        thr = float(rng.choice([0.0, -0.5, -1e-3]))
        h32 = h.astype(np.float32)
        if not np.array_equal(nms(t(h32), thr)[1].cpu().numpy(), oracle.hbb_nms(h32, thr)):
            bad.append(('f32', it, n, thr))
        if it % 2 == 0 and not np.array_equal(nms(h, thr)[1], oracle.hbb_nms_f64(h, thr)):
            bad.append(('f64', it, n, thr))
        if it % 5 == 0:
            grp = rng.integers(0, 6, n).astype(np.int32)
            if not np.array_equal(nms(t(h32), thr, groups=grp)[1].cpu().numpy(), oracle.hbb_nms(h32, thr, groups=grp)):
                bad.append(('groups', it, n, thr))
    assert not bad, bad
