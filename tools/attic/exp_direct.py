"""Crossover of the direct (unsorted) and the sorted NMS route over n."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import _lib
L = _lib.lib()

def lat_c(d, reps=50):
    n = d.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, 0), dtype=torch.uint8, device='cuda')
    info = (ctypes.c_int64 * 4)()
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        assert rc == 0, L.obb_last_error()
    for _ in range(5): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, info[0], info[3]

for per_box in (63.0, 13.0, 5.0):
    for n in (4096, 8192, 12000, 16384, 20000, 25000, 32768, 50000):
        canvas = (n ** 0.5) * per_box
        d = torch.from_numpy(oracle.synth_boxes(n, canvas, 5)).cuda()
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 65536)
        a, k, c = lat_c(d)
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
        b, k2, c2 = lat_c(d)
        assert k == k2
        print('density %4.0f n=%6d kept=%6d cand=%8d  direct %7.1f us   sorted %7.1f us' % (per_box, n, k, c, a, b))
