"""One-off randomised parity sweep of rbbox_overlaps and RRoIAlign (forward / backward) against
the oracle.  usage: python tools/fuzz_ops.py [iterations] [seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rbbox_overlaps, rroi_align
from aidet_amd.rbbox_geometry import pointobb_overlaps

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    # ---- overlaps
    m = int(np.exp(rng.uniform(0, np.log(1500)))); n = int(np.exp(rng.uniform(0, np.log(3000))))
    canvas = float(rng.choice([40, 400, 4000]))
    a = oracle.synth_boxes(m, canvas, 10 * it + 1)[:, :5]; b = oracle.synth_boxes(n, canvas, 10 * it + 2)[:, :5]
    a[:, :2] += rng.choice([0.0, 1e5]); b[:, :2] += a[0, 0] - a[0, 0]  # same offset for both sets
    if rng.random() < 0.3: b[: min(m, n)] = a[: min(m, n)]     # identical boxes -> IoU 1
    if rng.random() < 0.2: a[rng.integers(0, m), 2] = 0.0       # degenerate
    mode = 'iou' if it % 2 else 'iof'
    got = rbbox_overlaps(t(a), t(b), mode=mode).cpu().numpy()
    ref = oracle.overlaps(a, b, mode)
    if not np.array_equal(got, ref):
        bad += 1; print('overlaps MISMATCH it=%d m=%d n=%d max diff %.3g' % (it, m, n, np.abs(got - ref).max()))
    k = min(m, n)
    ga = rbbox_overlaps(t(a[:k]), t(b[:k]), mode=mode, is_aligned=True).cpu().numpy()
    if not np.array_equal(ga, np.diagonal(ref[:k, :k])):
        bad += 1; print('aligned MISMATCH it=%d' % it)
    if it % 4 == 0:
        pa, pb = oracle.rbox_to_poly(a).astype(np.float32), oracle.rbox_to_poly(b).astype(np.float32)
        gp = pointobb_overlaps(t(pa), t(pb), mode=mode).cpu().numpy()
        if not np.array_equal(gp, oracle.overlaps(pa, pb, mode, kind='poly')):
            bad += 1; print('poly overlaps MISMATCH it=%d' % it)
    # ---- rroi align
    B = int(rng.integers(1, 3)); C = int(rng.choice([1, 3, 8, 24, 40])); H = int(rng.integers(8, 90)); W = int(rng.integers(8, 90))
    K = int(rng.integers(1, 40)); ph = int(rng.integers(1, 9)); pw = int(rng.integers(1, 9)); sn = int(rng.choice([0, 1, 2, 3]))
    scale = float(rng.choice([1.0, 0.5, 0.25, 0.125]))
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    r = oracle.synth_boxes(K, max(W, H) / scale, 10 * it + 3)[:, :5]
    r[:, 2:4] *= rng.choice([0.2, 1.0, 3.0])
    rois = np.concatenate([rng.integers(0, B, (K, 1)).astype(np.float32), r], 1).astype(np.float32)
    ft = t(feat).requires_grad_(True)
    out = rroi_align(ft, t(rois), (ph, pw), scale, sn)
    ref = oracle.rroi_align_forward(feat, rois, (ph, pw), scale, sn)
    if np.abs(out.detach().cpu().numpy() - ref).max() > 1e-4 * max(1.0, np.abs(ref).max()):
        bad += 1; print('rroi fwd MISMATCH it=%d diff %.3g' % (it, np.abs(out.detach().cpu().numpy() - ref).max()))
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(t(g))
    rb = oracle.rroi_align_backward(g, rois, feat.shape, scale, sn)
    if np.abs(ft.grad.cpu().numpy() - rb).max() > 1e-4 * max(1.0, np.abs(rb).max()):
        bad += 1; print('rroi bwd MISMATCH it=%d diff %.3g (ref max %.3g)' % (it, np.abs(ft.grad.cpu().numpy() - rb).max(), np.abs(rb).max()))
print('fuzz_ops: %d iterations, %d mismatches' % (iters, bad))
sys.exit(1 if bad else 0)
