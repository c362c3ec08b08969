"""Soak of `soft_nms_grouped` (position-free kernel per group, given-up groups on the array kernel behind it) against the
oracle's loop per group: random group counts and sizes (0 .. 4500 rows), ties / NaN / -0 in some groups, both methods.
usage: python tools/soak_soft_grouped.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import nms_wrapper

a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
for it in range(a, b):
    rng = np.random.default_rng(52000 + it)
    G = int(rng.integers(1, 40))
    top = float(rng.choice([60, 400, 1500, 4500]))
    sizes = [int(np.exp(rng.uniform(0, np.log(top)))) - 1 for _ in range(G)]
    canvas = float(rng.choice([80, 300, 1200]))
    groups = []
    for gi, m in enumerate(sizes):
        x1, y1 = rng.uniform(0, canvas, m), rng.uniform(0, canvas, m)
        g = np.stack([x1, y1, x1 + rng.uniform(2, 100, m), y1 + rng.uniform(2, 100, m), rng.random(m)], 1).astype(np.float32)
        if m > 5 and rng.random() < 0.3:
            g[:, 4] = np.round(g[:, 4], int(rng.integers(1, 3)))
        if m > 5 and rng.random() < 0.1:
            g[rng.integers(0, m), 4] = np.nan
        if m > 5 and rng.random() < 0.1:
            g[rng.integers(0, m), 4] = -0.0
        groups.append(g)
    rows = torch.from_numpy(np.concatenate(groups, 0) if sum(sizes) else np.zeros((0, 5), np.float32)).cuda()
    min_score = float(rng.choice([1e-3, 0.05, 0.3]))
    thr = float(rng.choice([0.0, 0.3, 0.5]))
    for method in ('linear', 'gaussian'):
        res = nms_wrapper.soft_nms_grouped(rows, sizes, thr, method=method, sigma=0.5, min_score=min_score)
        for g, (d, k) in zip(groups, res):
            rd, ri = oracle.soft_nms(g, thr, method, 0.5, min_score)
            assert np.array_equal(k.cpu().numpy(), ri) and np.array_equal(d.cpu().numpy(), rd, equal_nan=True), (it, method, len(g))
    if it % 20 == 0:
        print('set', it, 'groups', G, 'rows', sum(sizes), 'ok  %.0f s' % (time.time() - t0), flush=True)
print('soak ok: sets %d..%d' % (a, b - 1))
