"""Soft-NMS of a detection head: 15 classes in ONE launch (soft_nms_grouped) vs one call per class, at the shape of
tests/golden/multiclass_soft.npz scaled up (2000 RoIs, score_thr 0.05: a few hundred candidates per class)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidet_amd.nms_wrapper import soft_nms, soft_nms_grouped
rng = np.random.default_rng(4)
dev = torch.device('cuda', 0)
for per in (150, 400, 1000):
    parts, sizes = [], []
    for c in range(15):
        m = int(rng.integers(per // 2, per + 1))
        x1, y1 = rng.uniform(0, 900, m), rng.uniform(0, 900, m)
        parts.append(np.stack([x1, y1, x1 + rng.uniform(8, 200, m), y1 + rng.uniform(8, 200, m), rng.uniform(0.05, 1, m)], 1).astype(np.float32))
        sizes.append(m)
    d = torch.from_numpy(np.concatenate(parts)).to(dev)
    def grouped():
        return soft_nms_grouped(d, sizes, 0.5, min_score=0.05)
    def loop():
        o, res = 0, []
        for m in sizes:
            res.append(soft_nms(d[o:o + m], 0.5, min_score=0.05)); o += m
        return res
    for f in (grouped, loop):
        f(); torch.cuda.synchronize()
    ts = {}
    for name, f in (('grouped', grouped), ('per class', loop)):
        t0 = time.perf_counter()
        for _ in range(5):
            r = f()
        torch.cuda.synchronize()
        ts[name] = (time.perf_counter() - t0) / 5 * 1e3
    a, b = grouped(), loop()
    same = all(torch.equal(x[1], y[1]) and torch.equal(x[0], y[0]) for x, y in zip(a, b))
    print('15 classes x %4d..%4d boxes: grouped %.3f ms   one call per class %.3f ms   identical: %s' % (per // 2, per, ts['grouped'], ts['per class'], same), flush=True)
