"""rnms latency when a few boxes lie far outside the scene (they stretch the quantisation frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms

def lat(d, reps=20):
    for _ in range(3): rnms(d, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): k = rnms(d, 0.5)[1]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, k.numel()

for n, canvas in ((100000, 20000), (20000, 8900)):
    base = oracle.synth_boxes(n, canvas, 3)
    for tag, far, cnt in (('none', 0, 0), ('1 box at 1e6', 1e6, 1), ('1 box at 1e8', 1e8, 1), ('10 boxes at 1e7', 1e7, 10), ('1% at 1e6', 1e6, n // 100)):
        d = base.copy()
        if cnt:
            idx = np.arange(cnt) * (n // cnt)
            d[idx, 0] += far
            d[idx, 1] -= far * 0.5
        us, k = lat(torch.from_numpy(d).cuda())
        print('n=%6d outliers: %-18s %8.1f us  kept %d' % (n, tag, us, k))
