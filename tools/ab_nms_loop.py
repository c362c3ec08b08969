"""A/B of two builds of the library on the headline call: OBBOPS_LIB names the library (both sides through the ctypes path:
tools/attic/ab_libs.sh explains why), a pool of four scenes, synchronous calls on the wrapper's retained workspace.
usage: OBBOPS_LIB=<lib> python tools/ab_nms_loop.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
if len(sys.argv) > 2:      # knob=value pairs, e.g. 22=0 (OBB_TUNE_NMS_PROBE_PAIR off)
    for kv in sys.argv[2:]:
        k, v = kv.split('=')
        _lib.set_tuning(int(k), int(v))
dev = torch.device('cuda', 0)
pool = [torch.from_numpy(workloads.boxes(n, 20000.0 * (n / 100000.0) ** 0.5, 3 + 10 * k)).to(dev) for k in range(4)]
for i in range(20):
    _run_nms(pool[i & 3], _lib.KIND_RBOX, 0.5)
res = []
for rep in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        _run_nms(pool[i & 3], _lib.KIND_RBOX, 0.5)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 200 * 1e6)
print('%-30s %s n %d: us per call (7 x 200 calls) %s  median %.2f  retained %d' % (os.path.basename(os.environ.get('OBBOPS_LIB', 'product')), ' '.join(sys.argv[2:]), n,
      ' '.join('%.1f' % x for x in res), float(np.median(res)), _lib.lib().obb_nms_last_retained()))
