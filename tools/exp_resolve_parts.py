"""Workgroups per edge shard in the resolve of a large call (OBB_TUNE_NMS_RESOLVE_PARTS): the merged NMS of a config-5 pass.
  python3 tools/exp_resolve_parts.py [images]"""
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
ref = None
for parts, mode in ((1, 1), (2, 1), (4, 1), (6, 1), (8, 1), (16, 1), (4, 0), (4, 3), (4, 2), (1, 2), (-1, -1)):     # (parts above what the device keeps resident / 2 are clamped by the library)
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, parts)
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)     # bit 0: register tail, bit 1: first sweep with agent-scope loads like the others
    for _ in range(3):
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.3, grp, thr)
    k = keep.cpu().numpy()
    assert ref is None or np.array_equal(ref, k)
    ref = k
    L.obb_nms_profile_enable(2)
    tab = np.zeros(7); buf = (ctypes.c_float * 7)()
    for _ in range(5):
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.3, grp, thr)
        L.obb_nms_profile_last(buf); tab += np.frombuffer(buf, dtype=np.float32)
    L.obb_nms_profile_enable(0)
    print('parts %2d mode %d: rows %d kept %d edges %d sweeps %d; resolve phase %.1f us (whole call phases %s)' % (parts, mode, K, keep.numel(), info[0], info[1], tab[5] / 5 * 1e3, (tab / 5 * 1e3).round(1).tolist()), flush=True)
_lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, -1)
_lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, -1)
