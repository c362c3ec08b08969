"""rnms latency when a few boxes are as large as the scene (every tile meets theirs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms
def lat(d, reps=10):
    for _ in range(2): rnms(d, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): k = rnms(d, 0.5)[1]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, k.numel()
n = 100000
base = oracle.synth_boxes(n, 20000, 3)
for cnt in (0, 1, 10, 100, 1000):
    d = base.copy()
    if cnt:
        idx = np.arange(cnt) * (n // cnt)
        d[idx, 2] = 15000.0; d[idx, 3] = 12000.0
    us, k = lat(torch.from_numpy(d).cuda())
    print('giant boxes: %5d  %9.1f us  kept %d' % (cnt, us, k))
