"""Soft-NMS on one array: the GPU kernels against the host loop (the oracle's restatement of nms_cpu.cpp:71-210) --
python3 tools/bench_soft_nms.py.  Column `gpu`: the default dispatch (one wave up to 64 boxes, the workgroup kernel above); `one wave`: the one-wave
kernel forced up to 512 boxes (shape 4); `workgroup`: never the one-wave kernel (shape 3).  Boxes of 8-200 px on a 900 px canvas; medians of 30 calls, host: best of 5."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import oracle
from aidet_amd import _lib
from aidet_amd.nms_wrapper import soft_nms

rng = np.random.default_rng(2)
for n in (64, 100, 200, 300, 500, 700, 1000, 2000, 4096):
    x1, y1 = rng.uniform(0, 900, n), rng.uniform(0, 900, n)
    hb = np.stack([x1, y1, x1 + rng.uniform(8, 200, n), y1 + rng.uniform(8, 200, n), (rng.permutation(n) + .5) / n], 1).astype(np.float32)
    hd = torch.from_numpy(hb).cuda()
    for method in ('linear', 'gaussian'):
        row = {}
        for shape in (0, 3, 4):
            _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, shape)
            ts = []
            for _ in range(34):
                t0 = time.perf_counter(); sd, si = soft_nms(hd, 0.5, method=method, min_score=0.05); ts.append(time.perf_counter() - t0)
            row[shape] = float(np.median(ts[4:])) * 1e3
        _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, -1)
        tcs = []
        for _ in range(5):
            t0 = time.perf_counter(); rd, ri = oracle.soft_nms(hb, 0.5, method, 0.5, 0.05); tcs.append(time.perf_counter() - t0)
        cpu = min(tcs) * 1e3
        same = np.array_equal(si.cpu().numpy(), ri) and np.array_equal(sd.cpu().numpy(), rd)
        print('n %5d %-8s gpu %.3f ms (workgroup %.3f, one wave %s) | host loop %.3f ms | x%.2f | kept %d, bit- and index-equal: %s'
              % (n, method, row[0], row[3], ('%.3f' % row[4]) if n <= 512 else '-', cpu, cpu / row[0], len(ri), same), flush=True)
