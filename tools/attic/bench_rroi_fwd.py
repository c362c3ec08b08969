"""Stand-alone RRoIAlign forward loop for rocprofv3 passes: python3 tools/bench_rroi_fwd.py [given|morton] [groups]
(configs[2]: 512 rotated RoIs x 256 ch on a 256^2 map; `morton` = the same RoIs sorted by the Morton code of their
centre on the host; groups = channels per block / 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib, RRoIAlign
order = sys.argv[1] if len(sys.argv) > 1 else 'given'
groups = int(sys.argv[2]) if len(sys.argv) > 2 else -1
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
if order == 'morton':
    def spread(v):
        v = (v | (v << 4)) & 0x0F0F; v = (v | (v << 2)) & 0x3333; v = (v | (v << 1)) & 0x5555
        return v
    r = rois.cpu().numpy()
    cx = (r[:, 1] / 4).astype(np.int64).clip(0, 255); cy = (r[:, 2] / 4).astype(np.int64).clip(0, 255)
    o = np.argsort(spread(cx) | (spread(cy) << 1), kind='stable')
    rois = rois[torch.from_numpy(o).to(dev)].contiguous()
_lib.set_tuning(_lib.TUNE_RROI_GROUPS, groups)
layer = RRoIAlign(7, 0.25, sample_num=2)
for _ in range(3):
    layer(feat, rois)
print(order, groups, 'ms', bench._time_gpu(lambda: layer(feat, rois), 10))
