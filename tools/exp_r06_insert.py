"""Round 6: what would one atomic + one store per box buy the grid route's insert launch?  Phase times (HIP events, profile
level 2) of the product library against an ablation build that registers every box in its lower cell only (WRONG results:
timing only; -DOBB_GINS_ABL=4, tools/exp_r06_insert.py builds nothing -- see the command in DESIGN_LOG).
usage: OBBOPS_LIB=<lib> python tools/exp_r06_insert.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
L = _lib.lib()
import ctypes
def phases(d, groups=None, reps=30):
    out = []
    for _ in range(5): _run_nms(d, _lib.KIND_RBOX, 0.5, groups)
    L.obb_nms_profile_enable(2)
    for _ in range(reps):
        _run_nms(d, _lib.KIND_RBOX, 0.5, groups)
        ms = (ctypes.c_float * 7)()
        L.obb_nms_profile_last(ms)
        out.append(list(ms))
    L.obb_nms_profile_enable(0)
    return np.median(np.array(out), 0) * 1e3
print(os.environ.get('OBBOPS_LIB', 'product'))
for n, canvas in ((100000, 20000), (1250000, 70000)):
    d = torch.from_numpy(workloads.boxes(n, canvas, 3)).to(dev)
    print(n, 'ungrouped: phases us', np.round(phases(d), 1), 'route', L.obb_nms_last_route(), 'retained', L.obb_nms_last_retained())
    g = torch.from_numpy(np.random.default_rng(1).integers(0, 240, n).astype(np.int32)).to(dev)
    print(n, '240 groups: phases us', np.round(phases(d, g), 1), 'route', L.obb_nms_last_route())
