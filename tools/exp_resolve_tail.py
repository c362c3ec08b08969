"""The resolve kernel's register tail (OBB_TUNE_NMS_RESOLVE_TAIL) on and off: the headline call (100 000 rotated boxes),
a 20 000-box call, a long suppression chain, and the config-5 merged NMS's resolve phase.
  python3 tools/exp_resolve_tail.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
L = _lib.lib()
cases = {}
for n, canvas in ((100000, 20000), (20000, 8000), (2048, 2048)):
    cases['%d boxes' % n] = torch.from_numpy(workloads.boxes(n, canvas, 3)).to(dev)
# a chain: box k overlaps k-1 and k+1 only, descending scores along it (every sweep retires one link)
m = 3000
ch = np.zeros((m, 6), np.float32)
ch[:, 0] = 10.0 * np.arange(m); ch[:, 1] = 50.0; ch[:, 2] = 16.0; ch[:, 3] = 16.0; ch[:, 5] = 1.0 - np.arange(m) / (2.0 * m)
cases['chain of %d' % m] = torch.from_numpy(ch).to(dev)
ref = {}
for tail in (0, 1, 0, 1):
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, tail)
    for name, d in cases.items():
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.5 if 'chain' not in name else 0.1)
        k = keep.cpu().numpy()
        if name in ref:
            assert np.array_equal(ref[name], k), name
        ref[name] = k
        ms = bench._time_gpu(lambda: _run_nms(d, _lib.KIND_RBOX, 0.5 if 'chain' not in name else 0.1), 30)
        L.obb_nms_profile_enable(2)
        buf = (ctypes.c_float * 7)(); tab = np.zeros(7)
        for _ in range(5):
            _run_nms(d, _lib.KIND_RBOX, 0.5 if 'chain' not in name else 0.1)
            L.obb_nms_profile_last(buf); tab += np.frombuffer(buf, dtype=np.float32)
        L.obb_nms_profile_enable(0)
        print('tail %d  %-16s call %.4f ms  kept %d edges %d rounds %d  resolve phase %.1f us' % (tail, name, ms, keep.numel(), info[0], info[1], tab[5] / 5 * 1e3), flush=True)
_lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, -1)
