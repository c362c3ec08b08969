"""The two NMS calls of a config-5 pass (per-patch: group = patch x class from int64 arrays; merged: group = image x
class and per-class thresholds from row columns) on the grid route and on the sorted route: route taken (from the
record), time per call, kernels in between are the tiled phases' own.  python tools/exp_tiled_grid.py [images]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch

import workloads
from aidet_amd import _lib, dota_io, dota_merge

dev = torch.device('cuda', 0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = workloads.tiled_scene(images=images)
C = scene['num_classes']
args = [torch.from_numpy(scene[k]).to(dev) for k in ('dets', 'label', 'patch')]
patch_xy = torch.from_numpy(scene['patch_xy']).to(dev)
patch_image = torch.from_numpy(scene['patch_image']).to(dev)
thrs = dota_io.class_thresholds()
n = args[0].size(0)
print('%d images, %d detections' % (images, n))


def run(grid):
    _lib.set_tuning(_lib.TUNE_NMS_GRID, grid)
    out = {}
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows, counts = dota_merge.tiled_local_phase(*args, patch_xy, patch_image, C, 0.5, 1)
        c = counts.tolist()
        t1 = time.perf_counter()
        K = c[2]
        route = c[3 + 6] >> 32 & 0xFFFFFFFF if False else None
        rec = np.asarray(c[3:], dtype=np.int64).view(np.uint32)
        out['local'] = (K, int(rec[12]) & 0xFF, (t1 - t0) * 1e6)
        if K < 0:
            info = (__import__('ctypes').c_int64 * 4)()
            nb = int(_lib.lib().obb_tiled_local_workspace_bytes(n, 1, 0))
            rc = _lib.lib().obb_tiled_result(np.asarray(c[3:], dtype=np.int64).ctypes.data, 0, n, 6, 1, -1, nb, info)
            print('  local K word', K, 'rc', rc)
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kept, keep, ordered = dota_merge.tiled_merge_phase(rows[:K], [0], None, K, C, 0.3, thrs, cols=8)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out['merge'] = (kept.size(0), (t1 - t0) * 1e6)
    return out


for grid in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, -1)):
    print('grid knob', grid, run(grid))
_lib.set_tuning(_lib.TUNE_NMS_GRID, -1)
