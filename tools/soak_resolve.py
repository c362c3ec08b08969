"""One-off soak of the resolve kernel's round-5 paths: random scenes of 2e3 .. 1.5e6 boxes with MANY overlap edges (dense
clusters, chains of overlapping boxes: long suppression chains), the register tail on / off, the cached first sweep on / off,
1 / 4 / 8 workgroups per edge shard -- every combination must return the keep list of the plain sweeps (and, up to 20 000
boxes, the oracle's).  usage: python tools/soak_resolve.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, workloads
from aidet_amd import _lib, rnms

a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
for it in range(a, b):
    rng = np.random.default_rng(515000 + it)
    n = int(np.exp(rng.uniform(np.log(2e3), np.log(1.5e6 if it % 3 else 6e4))))
    per_box = float(np.exp(rng.uniform(np.log(3.0 if n < 60000 else 7.0), np.log(30.0))))       # canvas side per sqrt(n): very dense .. moderate
    d = workloads.boxes(n, (n ** 0.5) * per_box, it)
    if it % 4 == 0:      # chains: runs of boxes a third of a width apart, scores falling along the run
        L = int(rng.choice([50, 400, 3000]))
        m = min(n // 2, 200 * L)
        k = np.arange(m)
        run, pos = k // L, k % L
        d[:m, 0] = (run % 64) * 400.0 + pos * (0.35 * 16.0) * (1 + (run % 3))
        d[:m, 1] = (run // 64) * 60.0 + 1e4
        d[:m, 2] = 16.0; d[:m, 3] = 16.0; d[:m, 4] = 0.0
        d[:m, 5] = 0.99 - pos / (2.0 * L) - run * 1e-7
    if it % 5 == 1:
        m = n // 3
        d[:m, :2] = d[:m, :2] * 0.05 + d[0, :2]
    if it % 6 == 2:
        d[:, 5] = np.round(d[:, 5], 3)
    dt = torch.from_numpy(d).cuda()
    groups = None
    if it % 3 == 0:
        ng = int(rng.choice([2, 40, 3000]))
        groups = torch.from_numpy(rng.integers(0, ng, n).astype(np.int32)).cuda()
    thr = float(rng.choice([0.1, 0.3, 0.5]))
    ref = None
    info0 = None
    for mode, parts in ((2, 1), (1, 4), (0, 4), (3, 8), (1, 1), (1, 8), (1, 2)):
        _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)
        _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, parts)
        try:
            keep = rnms(dt, thr, groups=groups)[1]
        except _lib.ObbOpsError as e:          # (a scene of almost all mutual overlaps: more pairs than the workspace budget)
            assert 'pair workspace' in str(e), e
            ref = None
            break
        if ref is None:
            ref = keep
        assert torch.equal(keep, ref), (it, mode, parts, n)
    if ref is None:
        continue
    if n <= 20000 and groups is None:
        assert np.array_equal(ref.cpu().numpy(), oracle.rnms(d, thr)), (it, n)
    if it % 10 == 0:
        print('scene', it, 'n', n, 'kept', ref.numel(), 'route', _lib.lib().obb_nms_last_route(), 'ok  %.0f s' % (time.time() - t0), flush=True)
_lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, -1)
_lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, -1)
print('soak ok: scenes %d..%d' % (a, b - 1))
