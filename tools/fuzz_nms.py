"""One-off randomised parity sweep of rnms / poly_nms against the oracle over all three routes.
usage: python tools/fuzz_nms.py [iterations] [seed0] [n_min] [n_max]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms, poly_nms

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_min = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n_max = int(sys.argv[4]) if len(sys.argv) > 4 else 6000
bad = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    n = int(np.exp(rng.uniform(np.log(n_min), np.log(n_max))))
    k = int(rng.integers(1, 30))
    centres = rng.uniform(0, rng.choice([50, 500, 5000]), (k, 2)) + rng.choice([0.0, 1e4, -3e5])
    which = rng.integers(0, k, n)
    d = np.zeros((n, 6), np.float32)
    d[:, :2] = centres[which] + rng.normal(0, rng.uniform(1, 100), (n, 2))
    d[:, 2] = np.exp(rng.uniform(np.log(1), np.log(rng.choice([30, 300])), n))
    d[:, 3] = np.exp(rng.uniform(np.log(1), np.log(rng.choice([30, 100])), n))
    d[:, 4] = rng.uniform(-np.pi, np.pi, n) if it % 3 else rng.choice([0.0, -np.pi / 2, np.pi / 4], n)
    d[:, 5] = rng.random(n)
    if it % 2: d[:, 5] = np.round(d[:, 5], int(rng.integers(1, 3)))
    if it % 5 == 0 and n > 10:   # duplicates and outliers
        d[n // 2:n // 2 + n // 10] = d[:n // 10]
        d[rng.integers(0, n), 0] += 1e7
    thr = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95]))
    groups = None
    if it % 4 == 0:
        ng = int(rng.choice([2, 3, 15, 400, 5000]))
        ids = rng.integers(-2**31, 2**31 - 1, ng, dtype=np.int64).astype(np.int32)
        groups = ids[rng.integers(0, ng, n)]
    thrs = rng.choice([0.1, 0.3, 0.5, 0.7], n).astype(np.float32) if it % 6 == 0 else None
    ref = oracle.rnms(d, thr, groups=groups, thrs=thrs)
    dt = torch.from_numpy(d).cuda()
    for route, env in (('auto', {}), ('sorted', {'OBB_NMS_DIRECT_MAX': '0', 'OBB_NMS_NO_FUSED': '1'}), ('direct', {'OBB_NMS_NO_FUSED': '1'}),
                       ('fused', {'OBB_NMS_FUSED_MAX': '1024'})):
        for kk in ('OBB_NMS_DIRECT_MAX', 'OBB_NMS_NO_FUSED', 'OBB_NMS_FUSED_MAX'): os.environ.pop(kk, None)
        os.environ.update(env)
        got = rnms(dt, thr, groups=groups, thrs=thrs)[1].cpu().numpy()
        if not np.array_equal(got, ref):
            bad += 1
            print('MISMATCH it=%d route=%s n=%d thr=%.2f: %d vs %d kept' % (it, route, n, thr, len(got), len(ref)))
    if it % 7 == 0:
        p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
        for kk in ('OBB_NMS_DIRECT_MAX', 'OBB_NMS_NO_FUSED', 'OBB_NMS_FUSED_MAX'): os.environ.pop(kk, None)
        if not np.array_equal(poly_nms(torch.from_numpy(p).cuda(), thr)[1].cpu().numpy(), oracle.poly_nms(p, thr)):
            bad += 1
            print('MISMATCH poly it=%d n=%d' % (it, n))
print('fuzz: %d iterations, %d mismatches' % (iters, bad))
sys.exit(1 if bad else 0)
