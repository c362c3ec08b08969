"""rbbox_overlaps 16k x 16k with / without the separating-axis filter, at configs[0]'s density, the sparse and the all-overlap
extremes: python3 tools/exp_overlaps_sat.py   (interleaved rounds, median / min ms)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib, rbbox_overlaps
dev = torch.device('cuda', 0)
for tag, mk in (('canvas 1024 (configs[0] density)', lambda: bench.make_iou(dev, 16384, 1024)), ('canvas 8192 (sparse)', lambda: bench.make_iou(dev, 16384, 8192)),
                ('all pairs overlapping 4096^2', lambda: bench.make_iou_allclip(dev, 4096))):
    a, b = mk()
    res = {0: [], 1: []}
    for _ in range(7):
        for sat in (0, 1):
            _lib.set_tuning(_lib.TUNE_OVERLAPS_SAT, sat)
            res[sat].append(bench._time_gpu(lambda: rbbox_overlaps(a, b), 10))
    _lib.set_tuning(_lib.TUNE_OVERLAPS_SAT, -1)
    alg = 20 * (a.size(0) + b.size(0)) + 4 * a.size(0) * b.size(0)
    for sat in (0, 1):
        m = float(np.median(res[sat]))
        print('%-34s sat=%d  median %.4f ms  min %.4f ms  = %.3f of 8 TB/s' % (tag, sat, m, min(res[sat]), alg / (m * 1e-3) / 8e12), flush=True)
