"""rnms latency on small, very dense scenes (RPN-like): the resolve phase dominates there."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import _lib
L = _lib.lib()
def lat_c(d, thr, reps=50):
    n = d.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, 0), dtype=torch.uint8, device='cuda')
    info = (ctypes.c_int64 * 4)()
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, thr, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, L.obb_last_error()
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, info[0], info[1], info[2]
for n, canvas, thr in ((2048, 300, 0.5), (2048, 150, 0.7), (6000, 400, 0.7), (10000, 700, 0.5), (30000, 1000, 0.5)):
    d = torch.from_numpy(oracle.synth_boxes(n, canvas, 5)).cuda()
    us, k, e, r = lat_c(d, thr)
    print('n=%6d canvas=%5d thr=%.1f kept=%6d edges=%8d sweeps=%3d  %8.1f us' % (n, canvas, thr, k, e, r, us))
