"""Synchronous rnms calls of RPN size in a loop (direct route), for A/B runs of two libraries (OBBOPS_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
for n, canvas in ((500, 500), (2000, 1024), (5000, 1024), (8768, 1024), (10000, 4000)):
    d = torch.from_numpy(workloads.boxes(n, canvas, 3)).to(dev)
    fn = lambda: _run_nms(d, _lib.KIND_RBOX, 0.7)
    for _ in range(10): fn()
    r = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): keep, info = fn()
        torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 50 * 1e6)
    print('%6d boxes (canvas %d): kept %d edges %d route %d: %s us per call' % (n, canvas, keep.numel(), info[0], _lib.lib().obb_nms_last_route(), ' '.join('%.1f' % x for x in r)), flush=True)
