"""rroi_plan_kernel over the RoI size range (SIZES=lo,hi in image pixels; configs[2] is 8,128): run under
rocprofv3 --kernel-trace --stats to read the plan kernel's duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
lo, hi = [float(v) for v in os.environ.get('SIZES', '8,128').split(',')]
rng = np.random.default_rng(5)
K = 512
r = np.stack([np.zeros(K), rng.uniform(0, 1024, K), rng.uniform(0, 1024, K), np.exp(rng.uniform(np.log(lo), np.log(hi), K)),
              np.exp(rng.uniform(np.log(lo), np.log(hi), K)), rng.uniform(-np.pi / 2, 0, K)], 1).astype(np.float32)
rois = torch.from_numpy(r).to(dev)
gout = torch.randn(K, 256, 7, 7, device=dev)
g1 = torch.empty(1, 256, 256, 256, device=dev)
ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev)
st = _lib.current_stream_handle(dev)
def gather():
    _lib.check(L.obb_roi_backward_gather_f32(gout.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0,
                                             g1.data_ptr(), ws.data_ptr(), ws.numel(), st))
print('sizes %g..%g px: backward %.4f ms' % (lo, hi, bench._time_gpu(gather, 20)))
