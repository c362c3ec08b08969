"""rnms on the headline workload through the two host paths of nms_wrapper._run_nms, interleaved in one process:
ctypes (aidet_amd._lib) vs the ATen-level binding (obbops_torch.nms_ex).  Median / p10 of per-call wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib, nms_wrapper
dev = torch.device('cuda', 0)
d = torch.from_numpy(workloads.boxes(100000, 20000.0, 3)).to(dev)
ext = nms_wrapper._fast_ext()
print('ext available:', ext is not None)
res = {'ctypes': [], 'ext': []}
for _ in range(20):
    nms_wrapper._run_nms(d, _lib.KIND_RBOX, 0.5)
for rep in range(300):
    for name in ('ctypes', 'ext'):
        nms_wrapper._ext = None if name == 'ctypes' else ext
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nms_wrapper._run_nms(d, _lib.KIND_RBOX, 0.5)
        res[name].append(time.perf_counter() - t0)
for name, ts in res.items():
    ts = np.array(ts[50:]) * 1e6
    print('%-7s median %.1f us  p10 %.1f  p90 %.1f' % (name, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
