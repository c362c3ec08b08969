"""RoI-Align forward / backward at configs[2] with the RoIs in their given order, sorted by footprint
size (largest first = LPT, smallest first) and in Morton order of their centres: is the kernel's time
set by a tail of large RoIs that start late?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
gout = torch.randn(512, 256, 7, 7, device=dev)
gfeat = torch.zeros_like(feat)
out = torch.empty(512, 256, 7, 7, device=dev)
st = _lib.current_stream_handle(dev)
r = rois.cpu().numpy()
w, h, th = r[:, 3] * .25, r[:, 4] * .25, r[:, 5]
fs = (np.floor(np.abs(np.cos(th)) * w + np.abs(np.sin(th)) * h) + 4) * (np.floor(np.abs(np.sin(th)) * w + np.abs(np.cos(th)) * h) + 4)
orders = {'given': np.arange(512), 'largest first': np.argsort(-fs), 'smallest first': np.argsort(fs)}
for name, o in orders.items():
    rr = rois[torch.from_numpy(o.copy()).to(dev)].contiguous()
    def fwd():
        _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, rr.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
    def bwd():
        _lib.check(L.obb_rroi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, rr.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, gfeat.data_ptr(), st))
    print('%-15s forward %.4f ms   backward %.4f ms' % (name, bench._time_gpu(fwd, 50), bench._time_gpu(bwd, 20)), flush=True)
