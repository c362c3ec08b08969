"""The merged NMS of a config-5 pass (1.25 M -> ~0.77 M rows, 240 (image, class) groups, per-class thresholds) as one
grouped rnms call: edges, resolve rounds, per-phase times (HIP events).  python3 tools/exp_cfg5_resolve.py [images]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib, dota_io, dota_merge
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = workloads.tiled_scene(images=images)
C = scene['num_classes']
args = [torch.from_numpy(scene[k]).to(dev) for k in ('dets', 'label', 'patch')]
rows, counts = dota_merge.tiled_local_phase(*args, torch.from_numpy(scene['patch_xy']).to(dev), torch.from_numpy(scene['patch_image']).to(dev), C, 0.5, 1)
K = counts.tolist()[2]
r = rows[:K]
d = r[:, :6].contiguous()
grp = (r[:, 7].to(torch.int64) * C + r[:, 6].to(torch.int64)).to(torch.int32)
thr = torch.tensor(dota_io.class_thresholds(), device=dev, dtype=torch.float32)[r[:, 6].to(torch.int64)]
L = _lib.lib()
for knob in (1, 0):
    _lib.set_tuning(_lib.TUNE_NMS_GRID, knob)
    for _ in range(3):
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.3, grp, thr)
    L.obb_nms_profile_enable(2)
    tab = np.zeros(7)
    buf = (ctypes.c_float * 7)()
    for _ in range(5):
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.3, grp, thr)
        L.obb_nms_profile_last(buf)
        tab += np.frombuffer(buf, dtype=np.float32)
    L.obb_nms_profile_enable(0)
    print('grid knob %d: rows %d kept %d edges %d resolve rounds %d candidates %d route %d' % (knob, K, keep.numel(), info[0], info[1], info[2], L.obb_nms_last_route()))
    print('   phases (us):', dict(zip(['frame/insert', 'cell_sort', 'permute_cull', 'probe/scan', 'clip', 'resolve', 'compact'], (tab / 5 * 1e3).round(1))))
_lib.set_tuning(_lib.TUNE_NMS_GRID, -1)
