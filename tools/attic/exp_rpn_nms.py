"""The RPN-shaped NMS of config 4 (5 levels x 2000 horizontal boxes in one grouped call, thr 0.7): direct
route vs the sorted routes, C-ABI latency."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms

rng = np.random.default_rng(2)
for per_level in (1000, 2000, 4000):
    n = 5 * per_level
    x1, y1 = rng.uniform(0, 900, n), rng.uniform(0, 900, n)
    hb = np.stack([x1, y1, x1 + rng.uniform(8, 200, n), y1 + rng.uniform(8, 200, n), rng.permutation(n) / n], 1).astype(np.float32)
    d = torch.from_numpy(hb).cuda()
    g = torch.arange(n, device='cuda', dtype=torch.int32) // per_level
    for name, direct, inline in (('auto', -1, -1), ('direct', 65536, -1), ('sorted in line', 0, 1 << 40), ('sorted listed', 0, 0)):
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, direct)
        _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, inline)
        for _ in range(5):
            k, st = _run_nms(d, _lib.KIND_HBB, 0.7, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            k, st = _run_nms(d, _lib.KIND_HBB, 0.7, g)
        torch.cuda.synchronize()
        print('n=%6d %-15s %7.1f us  kept %d edges %d' % (n, name, (time.perf_counter() - t0) / 50 * 1e6, k.numel(), st[0]), flush=True)
