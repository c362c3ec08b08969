"""RoI-Align: time forward / backward for several channels-per-block settings (tuning hook)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aidet_amd import _lib, RRoIAlign, RoIAlign
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
layer = RRoIAlign(7, 0.25, sample_num=2)
L = _lib.lib()
gout = torch.randn(512, 256, 7, 7, device=dev); gfeat = torch.zeros_like(feat)
st = _lib.current_stream_handle(dev)
def bwd():
    _lib.check(L.obb_rroi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, gfeat.data_ptr(), st))
for g in (1, 2, 4, 8):
    _lib.set_tuning(_lib.TUNE_RROI_GROUPS, g)
    f = bench._time_gpu(lambda: layer(feat, rois), 30)
    b = bench._time_gpu(bwd, 10)
    print(json.dumps({'chan_per_block': 8 * g, 'fwd_ms': round(f, 4), 'bwd_ms': round(b, 4)}))
# the config-4 shaped horizontal workload: 2000 RoIs of 8..200 px on the 256^2 level
import numpy as np
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
r = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1)
hrois = torch.from_numpy(r.astype(np.float32)).to(dev)
hl = RoIAlign(7, 0.25, sample_num=2)
# ... and the same RoIs in Morton order of their centres (what a spatial pre-sort would buy)
cx = ((r[:, 1] + r[:, 3]) / 2 / 4).astype(np.int64).clip(0, 255); cy = ((r[:, 2] + r[:, 4]) / 2 / 4).astype(np.int64).clip(0, 255)
def spread(v):
    v = (v | (v << 4)) & 0x0F0F; v = (v | (v << 2)) & 0x3333; v = (v | (v << 1)) & 0x5555
    return v
order = np.argsort(spread(cx) | (spread(cy) << 1), kind='stable')
srois = hrois[torch.from_numpy(order).to(dev)]
for g in (1, 2, 4, 8):
    _lib.set_tuning(_lib.TUNE_RROI_GROUPS, g)
    a = bench._time_gpu(lambda: hl(feat, hrois), 20)
    b = bench._time_gpu(lambda: hl(feat, srois), 20)
    print(json.dumps({'roi_fwd_2000': True, 'chan_per_block': 8 * g, 'ms': round(a, 4), 'ms_morton_sorted_rois': round(b, 4)}))
