"""Why is bench.py's timed loop slower than tools/exp_nms_watch.py's?  20 synchronous rnms calls between two synchronisations."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
L = _lib.lib()
d = torch.from_numpy(workloads.boxes(100000, 20000, 3)).to(dev)
step = lambda: _run_nms(d, _lib.KIND_RBOX, 0.5)
for _ in range(10): step()
def loop(K, probe_every):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        p = probe_every and i % probe_every == 0
        if p: L.obb_nms_profile_enable(1)
        step()
        if p: L.obb_nms_profile_enable(0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for watch in (1, 0):
    _lib.set_tuning(_lib.TUNE_NMS_WATCH_RECORD, watch)
    for K, pe in ((20, 0), (20, 4), (20, 10), (200, 0), (200, 4), (20, 1)):
        r = [loop(K, pe) for _ in range(5)]
        print('watch %d: %3d steps, probe every %2d: %s us per step' % (watch, K, pe, ' '.join('%.1f' % x for x in r)), flush=True)
