"""One-off soak beyond the oracle's reach: random scenes of 7e4 .. 1.2e6 boxes (sparse to dense, no / few /
very many groups, per-box thresholds, strict), the two sorted routes, the grid route (and the direct one below 65 536) must
return the same keep list.  usage: python tools/soak_routes.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib, rnms, poly_nms


def force(route):
    direct = {'direct': 65536, 'sorted': 0, 'listed': 0, 'grid': 0, 'auto': -1}[route]
    inline = {'direct': -1, 'sorted': 1 << 40, 'listed': 0, 'grid': -1, 'auto': -1}[route]
    grid = {'direct': 0, 'sorted': 0, 'listed': 0, 'grid': 1, 'auto': -1}[route]      # ('grid': wherever it applies; a scene it gives up is repeated on the sorted route)
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, direct)
    _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, inline)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, grid)


a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
for it in range(a, b):
    rng = np.random.default_rng(90000 + it)
    n = int(np.exp(rng.uniform(np.log(3e4), np.log(1.2e6))))
    per_box = float(np.exp(rng.uniform(np.log(6.0), np.log(80.0))))       # canvas side per sqrt(n): dense .. sparse
    d = workloads.boxes(n, (n ** 0.5) * per_box, it)
    if it % 5 == 0:      # a dense cluster holding a third of the boxes
        m = n // 3
        d[:m, :2] = d[:m, :2] * 0.05 + d[0, :2]
    if it % 4 == 1:
        d[:, 5] = np.round(d[:, 5], 3)       # tied scores
    nbig = 0
    if it % 4 == 2:      # a few boxes that touch more cells than an entry list names (the grid route's big list; more of them than it holds: the route gives up)
        k = int(rng.choice([1, 3, 40, 400]))
        rows = rng.choice(n, k, replace=False)
        side = (n ** 0.5) * per_box
        d[rows, 2] = side * rng.uniform(0.02, 0.4, k).astype(np.float32)
        d[rows, 3] = side * rng.uniform(0.02, 0.4, k).astype(np.float32)
        nbig = k
    dt = torch.from_numpy(d).cuda()
    groups = thrs = None
    if it % 3 == 0:
        ng = int(rng.choice([2, 40, 3000, 60000]))
        groups = torch.from_numpy(rng.integers(-ng, ng, n).astype(np.int32)).cuda()
    if it % 7 == 0:
        thrs = torch.from_numpy(rng.choice([0.1, 0.3, 0.5, 0.7], n).astype(np.float32)).cuda()
    thr, strict = float(rng.choice([0.3, 0.5, 0.7])), it % 6 == 0
    ref = None
    routes_taken = []
    for route in ('listed', 'sorted', 'direct', 'grid', 'auto'):
        if route == 'direct' and n > 65536:
            continue
        force(route)
        keep = rnms(dt, thr, groups=groups, thrs=thrs, strict=strict)[1]
        if ref is None:
            ref = keep
        assert torch.equal(keep, ref), (it, route, n)
        routes_taken.append(_lib.lib().obb_nms_last_route())
    if it % 10 == 0 or it % 4 == 2:
        print('scene', it, 'n', n, 'huge boxes', nbig, 'kept', ref.numel(), 'routes', routes_taken, 'ok  %.0f s' % (time.time() - t0), flush=True)
force('auto')
print('soak ok: scenes %d..%d' % (a, b - 1))
