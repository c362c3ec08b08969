"""RoI-Align forward at configs[2] (512 rotated RoIs) and at the config-4 shape (2000 horizontal RoIs) with the
RoIs in their given order vs. in Morton order of their centres, for 16 / 32 channels per block: how much of the
2.4x traffic is re-fetching that L2 locality removes?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib, RRoIAlign, RoIAlign
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)


def spread(v):
    v = (v | (v << 4)) & 0x0F0F; v = (v | (v << 2)) & 0x3333; v = (v | (v << 1)) & 0x5555
    return v


def morton(cx, cy):
    cx = (cx / 4).astype(np.int64).clip(0, 255); cy = (cy / 4).astype(np.int64).clip(0, 255)
    return np.argsort(spread(cx) | (spread(cy) << 1), kind='stable')


r = rois.cpu().numpy()
o = morton(r[:, 1], r[:, 2])
srois = rois[torch.from_numpy(o).to(dev)].contiguous()
layer = RRoIAlign(7, 0.25, sample_num=2)
for g in (2, 4):
    _lib.set_tuning(_lib.TUNE_RROI_GROUPS, g)
    a = bench._time_gpu(lambda: layer(feat, rois), 50)
    b = bench._time_gpu(lambda: layer(feat, srois), 50)
    print(json.dumps({'rroi_fwd_512': True, 'chan_per_block': 8 * g, 'ms': round(a, 4), 'ms_morton': round(b, 4)}), flush=True)
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
hr = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)
hrois = torch.from_numpy(hr).to(dev)
o = morton((hr[:, 1] + hr[:, 3]) / 2, (hr[:, 2] + hr[:, 4]) / 2)
shrois = hrois[torch.from_numpy(o).to(dev)].contiguous()
hl = RoIAlign(7, 0.25, sample_num=2)
for g in (2, 4):
    _lib.set_tuning(_lib.TUNE_RROI_GROUPS, g)
    a = bench._time_gpu(lambda: hl(feat, hrois), 20)
    b = bench._time_gpu(lambda: hl(feat, shrois), 20)
    print(json.dumps({'roi_fwd_2000': True, 'chan_per_block': 8 * g, 'ms': round(a, 4), 'ms_morton': round(b, 4)}), flush=True)
