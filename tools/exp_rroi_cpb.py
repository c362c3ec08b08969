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
