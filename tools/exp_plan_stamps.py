"""Phase stamps of the RoI-Align backward's plan kernel (instrumented build OBB_PLAN_STAMPS -> aidet_amd/lib/libobbops_plst.so)
at configs[2] (512 rotated RoIs, 256 channels, 7 x 7, 2 x 2 samples): where a plan block's life goes.
  python3 tools/exp_plan_stamps.py [nhwc]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_plst.so')
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from aidet_amd import _lib
nhwc = len(sys.argv) > 1 and sys.argv[1] == 'nhwc'
dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
feat, rois = bench.make_rroi(dev)
go = torch.randn(512, 256, 7, 7, device=dev)
gm = torch.empty_like(feat)
if nhwc:
    ws = torch.empty(L.obb_roi_backward_nhwc_workspace_bytes(512, 256, 7, 7), dtype=torch.uint8, device=dev)
    run = lambda: _lib.check(L.obb_roi_backward_gather_nhwc_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 0, 7, 7, 0.25, 2, 1.0,
                                                                   gm.data_ptr(), ws.data_ptr(), ws.numel(), st))
else:
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(512), dtype=torch.uint8, device=dev)
    run = lambda: _lib.check(L.obb_roi_backward_gather_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 0, 7, 7, 0.25, 2, 1.0,
                                                              gm.data_ptr(), ws.data_ptr(), ws.numel(), st))
for _ in range(20):
    run()
torch.cuda.synchronize()
ms = bench._time_gpu(run, 50)
run(); torch.cuda.synchronize()
buf = np.zeros(12 * 4096, dtype=np.uint64)
L.obb_debug_plan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_plan_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(4096, 12).astype(np.int64)[:512]
ok = (s[:, 8] > s[:, 0]) & (s[:, 7] > 0)
s = s[ok]
print('%s backward (stamped build) %.1f us; %d planned blocks' % ('channels-last' if nhwc else 'NCHW', ms * 1e3, len(s)))
names = ['entry -> frame, footprint, tile counts', '-> sample table + barrier', '-> taps: merge, count + barrier', '-> scan of the counts', '-> fill + barrier',
         '-> short lists ranked + barrier', '-> long lists', '-> record, finish']
tot = s[:, 8] - s[:, 0]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print('%-42s median %6d  mean %6d  p90 %6d cycles (%.0f %% of the block)' % (nm, np.median(d), d.mean(), np.percentile(d, 90), 100.0 * d.sum() / tot.sum()))
print('block life: median %d, p90 %d, max %d shader cycles' % (np.median(tot), np.percentile(tot, 90), tot.max()))
t0, t1 = s[:, 10], s[:, 11]
base = t0.min()
print('real time (10 ns ticks): last entry %d, first exit %d, last exit %d; block life median %d ticks'
      % (t0.max() - base, t1.min() - base, t1.max() - base, np.median(t1 - t0)))
