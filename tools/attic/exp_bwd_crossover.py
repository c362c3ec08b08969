"""RoI-Align backward, gather vs scatter formulation over the number of RoIs (configs[2]-like rotated RoIs on a 256^2
map, 256 channels): where should roi_layers.GATHER_MIN_ROIS sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, workloads
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
st = _lib.current_stream_handle(dev)
g1 = torch.empty(1, 256, 256, 256, device=dev); g2 = torch.zeros_like(g1)
for K in (16, 32, 64, 128, 256, 512, 1024):
    r = workloads.boxes(K, 1024, 5)[:, :5]
    rois = torch.from_numpy(np.concatenate([np.zeros((K, 1), np.float32), r], 1)).to(dev)
    gout = torch.randn(K, 256, 7, 7, device=dev)
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev)
    def gather():
        _lib.check(L.obb_roi_backward_gather_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0, g1.data_ptr(), ws.data_ptr(), ws.numel(), st))
    def scatter():
        g2.zero_()
        _lib.check(L.obb_rroi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 7, 7, 0.25, 2, 1.0, g2.data_ptr(), st))
    print('K %5d  gather %.4f ms   scatter + zero fill %.4f ms' % (K, bench._time_gpu(gather, 20), bench._time_gpu(scatter, 20)), flush=True)
