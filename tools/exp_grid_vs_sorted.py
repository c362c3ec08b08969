"""Grid route against the sorted route over n (ungrouped scenes at the headline's density; grouped: 240 groups on one
canvas of a quarter of that area each): time per synchronous call through the Python wrapper, route taken."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms

L = _lib.lib()
for n in (12000, 20000, 50000, 100000, 200000, 500000, 1000000, 1400000, 3000000, 10000000):
    canvas = 20000.0 * (n / 100000.0) ** 0.5
    d = torch.from_numpy(workloads.boxes(n, canvas, 3)).cuda()
    g = torch.from_numpy(np.random.default_rng(1).integers(0, 240, n).astype(np.int32)).cuda()
    dg = torch.from_numpy(workloads.boxes(n, canvas / 8.0, 4)).cuda()      # 240 groups overlaid: 64 x the density, / 240 per group
    row = [n]
    for tag, dets, grp in (('ungrouped', d, None), ('240 groups', dg, g)):
        for knob in (1, 0):
            _lib.set_tuning(_lib.TUNE_NMS_GRID, knob)
            for _ in range(3):
                k, _i = _run_nms(dets, _lib.KIND_RBOX, 0.5, grp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10 if n <= 1400000 else 3
            for _ in range(reps):
                k, _i = _run_nms(dets, _lib.KIND_RBOX, 0.5, grp)
            torch.cuda.synchronize()
            row.append('%s %s route %d: %.0f us' % (tag, 'grid' if knob else 'sorted', L.obb_nms_last_route(), (time.perf_counter() - t0) / reps * 1e6))
    print(' | '.join(str(r) for r in row), flush=True)
    del d, g, dg
    torch.cuda.empty_cache()
_lib.set_tuning(_lib.TUNE_NMS_GRID, -1)
