"""prepq of the merged NMS: rows read in place (NmsSource mode 2) vs pre-split boxes / groups / thresholds (mode 0),
and the same rows in a random order (hist contention of neighbouring rows).  Run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import dota_io, dota_merge
dev = torch.device('cuda', 0)
scene = workloads.tiled_scene()
C, P = scene['num_classes'], len(scene['patch_name'])
f = lambda a: torch.from_numpy(a).to(dev)
rows, counts = dota_merge.tiled_local_phase(f(scene['dets']), f(scene['label']), f(scene['patch']), f(scene['patch_xy']), f(scene['patch_image']), C, 0.5, 1)
K = int(counts[2])
rows = rows[:K].clone()
sub = rows[(rows[:, 7].long() % 8) == 0].contiguous()        # what rank 0 of 8 merges
thrs = dota_io.class_thresholds()
print('rows', sub.shape)
for tag, r in (('patch order', sub), ('random order', sub[torch.randperm(sub.size(0), device=dev)].contiguous())):
    for _ in range(3):
        dota_merge.tiled_merge_phase(r, [0], None, r.size(0), C, 0.3, thrs)           # mode 2
    torch.cuda.synchronize()
    for _ in range(3):
        dota_merge.merge_detections(r[:, :8], 0.3, C, thrs)                             # mode 0 (+ split kernel)
    torch.cuda.synchronize()
