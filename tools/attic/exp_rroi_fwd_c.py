"""RRoIAlign forward through the C ABI (no Python layer in the timed region): configs[2] and the config-4 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, oracle
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
st = _lib.current_stream_handle(dev)
K = rois.shape[0]
out = torch.empty(K, 256, 7, 7, device=dev)
def fwd():
    _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
for g in (-1, 2, 4, 8, 16):
    _lib.set_tuning(_lib.TUNE_RROI_GROUPS, g)
    ts = [bench._time_gpu(fwd, 30) for _ in range(5)]
    print('configs[2] forward (C ABI), %2d groups of 8 channels per block: best %.4f ms  median %.4f ms' % (g, min(ts), sorted(ts)[2]), flush=True)
_lib.set_tuning(_lib.TUNE_RROI_GROUPS, -1)
fwd(); torch.cuda.synchronize()
ref = oracle.rroi_align_forward(feat.cpu().numpy()[:, :8], rois.cpu().numpy(), 7, 0.25, 2)
print('max |diff| vs oracle (8 channels): %.3e' % float(np.abs(out.cpu().numpy()[:, :8] - ref).max()))
