"""Round 6 soak of the NMS on a retained workspace (OBB_NMS_RETAIN): random streams of scenes of 10 300 .. 262 000 boxes --
runs of one kind, then a jump in size class, place, scale or density, groups and per-box thresholds now and then -- through
the wrapper (the binding owns the workspace).  Every keep list must equal the sorted route's and the plain four-launch
call's.  usage: python tools/soak_retain.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib, rnms

L = _lib.lib()


def mode(grid, retain):
    _lib.set_tuning(_lib.TUNE_NMS_GRID, grid)
    _lib.set_tuning(_lib.TUNE_NMS_RETAIN, retain)


a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
seen = {0: 0, 1: 0, 2: 0}
routes = {}
for it in range(a, b):
    rng = np.random.default_rng(660000 + it)
    # a stream: one kind of scene for a few calls
    n0 = int(np.exp(rng.uniform(np.log(10300), np.log(262000))))
    per_box = float(np.exp(rng.uniform(np.log(20.0), np.log(90.0))))
    shift = rng.uniform(-1e5, 1e5, 2).astype(np.float32) if it % 3 == 0 else np.zeros(2, np.float32)
    scale = float(rng.choice([1.0, 1.0, 0.2, 4.0]))
    ng = int(rng.choice([0, 0, 3, 200]))
    stream = []
    for k in range(int(rng.integers(2, 6))):
        n = n0 if k % 2 == 0 else max(10300, min(262000, int(n0 * rng.uniform(0.8, 1.2))))
        d = workloads.boxes(n, (n ** 0.5) * per_box, 7 * it + k)
        d[:, :2] += shift
        d[:, 2:4] *= scale
        if (it + k) % 11 == 0:
            m = n // 4
            d[:m, :2] = d[:m, :2] * 0.03 + d[0, :2]       # a dense cluster
        dt = torch.from_numpy(d).cuda()
        groups = torch.from_numpy(rng.integers(0, ng, n).astype(np.int32)).cuda() if ng else None
        thrs = torch.from_numpy(rng.choice([0.2, 0.4, 0.6], n).astype(np.float32)).cuda() if (it + k) % 9 == 0 else None
        thr = float(rng.choice([0.3, 0.5]))
        mode(0, -1)
        ref = rnms(dt, thr, groups=groups, thrs=thrs)[1]
        mode(1, 0)
        plain = rnms(dt, thr, groups=groups, thrs=thrs)[1]
        assert torch.equal(plain, ref), ('plain', it, k, n)
        stream.append((dt, thr, groups, thrs, ref))
    # the stream on the retained workspace: every call finds what its predecessor (another scene) left; twice round
    mode(1, 1)
    for rep in range(2):
        for k, (dt, thr, groups, thrs, ref) in enumerate(stream):
            keep = rnms(dt, thr, groups=groups, thrs=thrs)[1]
            r = int(L.obb_nms_last_retained())
            seen[r] += 1
            routes[int(L.obb_nms_last_route())] = routes.get(int(L.obb_nms_last_route()), 0) + 1
            assert torch.equal(keep, ref), ('retained', it, k, rep, dt.size(0), r)
    if it % 10 == 0:
        print('stream', it, 'n', n0, 'retained states', seen, 'routes', routes, 'ok  %.0f s' % (time.time() - t0), flush=True)
mode(-1, -1)
print('soak ok: streams %d..%d, retained states %s, routes %s' % (a, b - 1, seen, routes))
