"""Per-phase shader-clock stamps of the RoI-Align forward kernel (instrumented build OBB_RROI_ABLATE=5 of
aidet_amd/csrc/obb_rroi.hip -> aidet_amd/lib/libobbops_abl5.so): where a block's life goes at configs[2]."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_abl5.so')
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from aidet_amd import _lib
dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
feat, rois = bench.make_rroi(dev)
out = torch.empty((512, 256, 7, 7), device=dev)
fwd = lambda: _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
for _ in range(20):
    fwd()
torch.cuda.synchronize()
ms = bench._time_gpu(fwd, 50)
fwd(); torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.uint64)
L.obb_debug_rroi_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_rroi_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(4096, 8).astype(np.int64)
ok = (s[:, 4] > s[:, 0]) & (s[:, 3] >= s[:, 2])
s = s[ok]
print('kernel %.1f us; %d blocks with a full set of stamps (the small-footprint path)' % (ms * 1e3, len(s)))
names = ['entry -> frame, footprint, copy offsets, first copy issued', '-> table built + barrier', '-> first copy landed (vmcnt(0))', '-> gathers + stores of all passes']
tot = s[:, 4] - s[:, 0]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print('%-62s median %6d  mean %6d  p90 %6d cycles (%.0f %% of the block)' % (nm, np.median(d), d.mean(), np.percentile(d, 90), 100.0 * d.sum() / tot.sum()))
print('block life: median %d, mean %d, p90 %d, max %d cycles (s_memtime ticks at the shader clock)' % (np.median(tot), tot.mean(), np.percentile(tot, 90), tot.max()))
print('4096 blocks / (256 CUs x 5 resident) = 3.2 rounds x %d cycles = %.0f k cycles = %.1f us at 2.1 GHz (the counters of the 8 XCDs are not synchronised: no global span)'
      % (np.median(tot), 3.2 * np.median(tot) / 1e3, 3.2 * np.median(tot) / 2.1e3))
