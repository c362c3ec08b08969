"""Per-call latency of rnms for small n: fused single-launch kernel vs general pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import _lib, rnms

def lat(d, reps=200):
    for _ in range(10): rnms(d, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): rnms(d, 0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for canvas_per_box in (30.0, 8.0):     # sparse-ish / dense (RPN-like)
    for n in (16, 64, 128, 256, 384, 512, 768, 1024):
        canvas = (n ** 0.5) * canvas_per_box * 3
        d = torch.from_numpy(oracle.synth_boxes(n, canvas, 5)).cuda()
        _lib.set_tuning(_lib.TUNE_NMS_FUSED_MAX, 1024)   # every size through the fused kernel
        f = lat(d)
        _lib.set_tuning(_lib.TUNE_NMS_FUSED_MAX, 0)      # ... and through the general routes
        g = lat(d)
        _lib.set_tuning(_lib.TUNE_NMS_FUSED_MAX, -1)
        print('n=%5d canvas=%6.0f  fused %7.1f us   general %7.1f us' % (n, canvas, f, g))
