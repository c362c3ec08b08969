"""RRoIAlign forward over the RoI size range (512 RoIs, 256 ch, 256^2 map): which RoIs cost what?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
feat = torch.randn(1, 256, 256, 256, device=dev)
st = _lib.current_stream_handle(dev)
K = 512
out = torch.empty(K, 256, 7, 7, device=dev)
for lo, hi in ((8, 128), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256)):
    rng = np.random.default_rng(5)
    r = np.stack([np.zeros(K), rng.uniform(0, 1024, K), rng.uniform(0, 1024, K), np.exp(rng.uniform(np.log(lo), np.log(hi), K)),
                  np.exp(rng.uniform(np.log(lo), np.log(hi), K)), rng.uniform(-np.pi / 2, 0, K)], 1).astype(np.float32)
    rois = torch.from_numpy(r).to(dev)
    def fwd():
        _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
    print('sizes %3d..%3d px: forward %.4f ms' % (lo, hi, min(bench._time_gpu(fwd, 30) for _ in range(3))), flush=True)
