"""The headline rnms call (100 000 boxes) with the host watching its pinned result record vs sleeping in
hipStreamSynchronize (OBB_TUNE_NMS_WATCH_RECORD 1 / 0): python3 tools/exp_nms_watch.py   -- interleaved rounds, us per call
through the Python wrapper and through the bare C ABI."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
L = _lib.lib()
for n, canvas in ((100000, 20000.0), (20000, 9000.0), (2048, 2900.0)):
    d = torch.from_numpy(workloads.boxes(n, canvas, 3)).to(dev)
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, 0), dtype=torch.uint8, device=dev)
    info = (ctypes.c_int64 * 4)()
    st = _lib.current_stream_handle(dev)
    res = {(w, how): [] for w in (0, 1) for how in ('python', 'c abi')}
    for _ in range(7):
        for w in (0, 1):
            _lib.set_tuning(_lib.TUNE_NMS_WATCH_RECORD, w)
            for how in ('python', 'c abi'):
                fn = (lambda: _run_nms(d, _lib.KIND_RBOX, 0.5)) if how == 'python' else \
                     (lambda: L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st))
                for _ in range(20):
                    fn()
                t0 = time.perf_counter()
                for _ in range(200):
                    fn()
                res[(w, how)].append((time.perf_counter() - t0) / 200 * 1e6)
    _lib.set_tuning(_lib.TUNE_NMS_WATCH_RECORD, -1)
    for (w, how), v in sorted(res.items()):
        print('n %6d  %-7s watch=%d  median %.1f us  min %.1f us per call' % (n, how, w, float(np.median(v)), min(v)), flush=True)
