"""Tail latency of rnms: per-call wall times over many calls (are there stalls?)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import _lib
L = _lib.lib()
for n in (100, 2048, 20000):
    d = torch.from_numpy(oracle.synth_boxes(n, (n ** 0.5) * 63.0, 7)).cuda()
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, 0), dtype=torch.uint8, device='cuda')
    info = (ctypes.c_int64 * 4)()
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        assert rc == 0
    for _ in range(20): call()
    ts = np.empty(20000); rs = np.empty(20000)
    for i in range(20000):
        t0 = time.perf_counter(); call(); ts[i] = time.perf_counter() - t0; rs[i] = info[2]
    ts *= 1e6
    print('n=%6d  median %.1f  p99 %.1f  p99.9 %.1f  max %.1f us   calls > 1 ms: %d' % (n, np.median(ts), np.percentile(ts, 99), np.percentile(ts, 99.9), ts.max(), (ts > 1000).sum()), ' sweeps of the slow calls:', rs[ts > 1000].tolist(), 'usual:', np.median(rs))
