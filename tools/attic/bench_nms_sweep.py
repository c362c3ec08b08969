"""Per-call latency of rnms over n (python wrapper included and C-ABI only)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms, _lib

L = _lib.lib()

def lat_py(d, reps=100):
    for _ in range(5): rnms(d, 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): rnms(d, 0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

def lat_c(d, reps=100):
    n = d.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, 0), dtype=torch.uint8, device='cuda')
    info = (ctypes.c_int64 * 4)()
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        assert rc == 0
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, info[0]

for per_box in (63.0, 13.0):     # canvas side per sqrt(n): sparse (headline density) / dense
    for n in (256, 512, 1024, 2048, 5000, 10000, 20000, 50000, 100000, 300000, 1000000):
        canvas = (n ** 0.5) * per_box
        d = torch.from_numpy(oracle.synth_boxes(n, canvas, 5)).cuda()
        c, k = lat_c(d, 100 if n <= 100000 else 20)
        p = lat_py(d, 100 if n <= 100000 else 20)
        print('n=%8d canvas=%7.0f kept=%8d  C-ABI %8.1f us   python %8.1f us   %7.1f Mboxes/s' % (n, canvas, k, c, p, n / c))
