"""Phase stamps of the channels-last gather backward (libobbops_nhst.so): list | filter | stream per tile block at configs[2]."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_nhst.so')
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from aidet_amd import _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
feat, rois = bench.make_rroi(dev, K=K)
go = torch.randn(K, 256, 7, 7, device=dev)
gl = torch.empty((1, 256, 256, 256), device=dev, memory_format=torch.channels_last)
wl = torch.empty(L.obb_roi_backward_nhwc_workspace_bytes(K, 256, 7, 7), dtype=torch.uint8, device=dev)
run = lambda: _lib.check(L.obb_roi_backward_gather_nhwc_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0,
                                                              gl.data_ptr(), wl.data_ptr(), wl.numel(), st))
for _ in range(10):
    run()
torch.cuda.synchronize()
ms = bench._time_gpu(run, 20)
run(); torch.cuda.synchronize()
buf = np.zeros(8 * 8192, dtype=np.uint64)
L.obb_debug_nhwc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_nhwc_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(8192, 8).astype(np.int64)[:1024]
print('K %d: whole call %.1f us (stamped build)' % (K, ms * 1e3))
life = s[:, 4] - s[:, 0]
for nm, d in (('entry -> list', s[:, 1] - s[:, 0]), ('filter (all batches)', s[:, 2]), ('stream (all batches)', s[:, 3]), ('block life', life)):
    print('%-24s median %7d  mean %7d  p90 %7d  max %7d cycles' % (nm, np.median(d), d.mean(), np.percentile(d, 90), d.max()))
print('hits per tile: median %d mean %d p90 %d max %d; batches: median %d max %d' % (np.median(s[:, 5]), s[:, 5].mean(), np.percentile(s[:, 5], 90), s[:, 5].max(), np.median(s[:, 6]), s[:, 6].max()))
i = int(np.argmax(life))
print('slowest block: life %d = list %d + filter %d + stream %d; hits %d, batches %d' % (life[i], s[i, 1] - s[i, 0], s[i, 2], s[i, 3], s[i, 5], s[i, 6]))
print('stream cycles per hit (all blocks): %.1f' % (s[:, 3].sum() / max(s[:, 5].sum(), 1)))
