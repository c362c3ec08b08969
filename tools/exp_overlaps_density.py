"""rbbox_overlaps 16k x 16k at configs[0]'s density, the sparse and the all-overlap extremes: python3 tools/exp_overlaps_density.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import rbbox_overlaps
dev = torch.device('cuda', 0)
for tag, mk in (('canvas 1024 (configs[0] density)', lambda: bench.make_iou(dev, 16384, 1024)), ('canvas 4096', lambda: bench.make_iou(dev, 16384, 4096)),
                ('canvas 8192 (sparse)', lambda: bench.make_iou(dev, 16384, 8192)), ('canvas 256 (dense)', lambda: bench.make_iou(dev, 8192, 256)),
                ('all pairs overlapping 4096^2', lambda: bench.make_iou_allclip(dev, 4096))):
    a, b = mk()
    res = [bench._time_gpu(lambda: rbbox_overlaps(a, b), 10) for _ in range(5)]
    alg = 20 * (a.size(0) + b.size(0)) + 4 * a.size(0) * b.size(0)
    m = float(np.median(res))
    print('%-34s median %.4f ms  min %.4f ms  = %.3f of 8 TB/s, %.3g pairs/s' % (tag, m, min(res), alg / (m * 1e-3) / 8e12, a.size(0) * b.size(0) / (m * 1e-3)), flush=True)
