"""Phase split (HIP events inside obb_nms_*) of the two NMS calls of config 5 on one GPU."""
import os, sys, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib, dota_io, dota_merge
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
sc = workloads.tiled_scene()
C = sc['num_classes']
dets = torch.from_numpy(sc['dets']).to(dev); labels = torch.from_numpy(sc['label']).to(dev); patch = torch.from_numpy(sc['patch']).to(dev)
L = _lib.lib()
names = ['prep_quant', 'cell_sort', 'permute_cull', 'scan', 'clip', 'resolve', 'compact']
def phases(fn, reps=5):
    fn(); L.obb_nms_profile_enable(2)
    acc = np.zeros(7); buf = (ctypes.c_float * 7)()
    for _ in range(reps):
        out = fn(); L.obb_nms_profile_last(buf); acc += np.frombuffer(buf, dtype=np.float32)
    L.obb_nms_profile_enable(0)
    return {k: round(float(v) / reps * 1e3, 1) for k, v in zip(names, acc)}, out
g1 = patch * C + labels
p1, (keep, info) = phases(lambda: _run_nms(dets, _lib.KIND_RBOX, 0.5, g1))
print(json.dumps({'call': 'local per-patch nms', 'n': int(dets.size(0)), 'kept': int(keep.numel()), 'edges': info[0], 'rounds': info[1], 'candidates': info[2], 'phase_us': p1}))
rows = dota_merge.patch_rows_to_image(dets[keep], labels[keep], patch[keep], torch.from_numpy(sc['patch_xy']).to(dev), torch.from_numpy(sc['patch_image']).to(dev))
thr = torch.tensor(dota_io.class_thresholds(), device=dev)[rows[:, 6].long()]
g2 = rows[:, 7].long() * C + rows[:, 6].long()
d2 = rows[:, :6].contiguous()
p2, (keep2, info2) = phases(lambda: _run_nms(d2, _lib.KIND_RBOX, 0.3, g2, thr))
print(json.dumps({'call': 'merged nms', 'n': int(d2.size(0)), 'kept': int(keep2.numel()), 'edges': info2[0], 'rounds': info2[1], 'candidates': info2[2], 'phase_us': p2}))
p3, (keep3, info3) = phases(lambda: _run_nms(d2, _lib.KIND_RBOX, 0.3, g2))
print(json.dumps({'call': 'merged nms, one threshold', 'kept': int(keep3.numel()), 'edges': info3[0], 'candidates': info3[2], 'phase_us': p3}))
p4, (keep4, info4) = phases(lambda: _run_nms(d2, _lib.KIND_RBOX, 0.3))
print(json.dumps({'call': 'same boxes, no groups', 'kept': int(keep4.numel()), 'edges': info4[0], 'candidates': info4[2], 'phase_us': p4}))
