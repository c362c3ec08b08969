"""Phase stamps of the channels-last RoI-Align forward (instrumented build OBB_NHWC_STAMPS -> aidet_amd/lib/libobbops_nhst.so):
where a block's life goes at configs[2], and when the blocks run (100 MHz real-time counter).
  python3 tools/exp_nhwc_stamps.py [nq] [waves]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_nhst.so')
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from aidet_amd import _lib
nq = int(sys.argv[1]) if len(sys.argv) > 1 else -1
nw = int(sys.argv[2]) if len(sys.argv) > 2 else -1
dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
feat, rois = bench.make_rroi(dev)
fl = feat.contiguous(memory_format=torch.channels_last)
out = torch.empty((512, 256, 7, 7), device=dev)
_lib.set_tuning(_lib.TUNE_RROI_NHWC_NQ, nq)
_lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, nw)
fwd = lambda: _lib.check(L.obb_rroi_align_forward_nhwc_f32(fl.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
for _ in range(20):
    fwd()
torch.cuda.synchronize()
ms = bench._time_gpu(fwd, 50)
fwd(); torch.cuda.synchronize()
buf = np.zeros(8 * 8192, dtype=np.uint64)
L.obb_debug_nhwc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_nhwc_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(8192, 8).astype(np.int64)
s = s[(s[:, 5] > s[:, 0]) & (s[:, 0] > 0)]
print('nq %d waves %d: kernel %.1f us (stamped build); %d blocks' % (nq, nw, ms * 1e3, len(s)))
names = ['entry -> frame (wave 0) + barrier', '-> sample table + barrier', "-> wave 0's gathers", '-> tile complete (barrier)', '-> copy-out issued']
tot = s[:, 5] - s[:, 0]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print('%-40s median %6d  mean %6d  p90 %6d cycles (%.0f %% of the block)' % (nm, np.median(d), d.mean(), np.percentile(d, 90), 100.0 * d.sum() / tot.sum()))
print('block life: median %d, p90 %d, max %d shader cycles' % (np.median(tot), np.percentile(tot, 90), tot.max()))
t0, t1 = s[:, 6], s[:, 7]
base = t0.min()
print('real time (10 ns ticks): first entry 0, last entry %d, first exit %d, last exit %d; block life median %d ticks -> clock %.2f GHz'
      % (t0.max() - base, t1.min() - base, t1.max() - base, np.median(t1 - t0), np.median(tot) / max(np.median(t1 - t0), 1) / 10.0))
h, _ = np.histogram(t0 - base, bins=8, range=(0, max(t1.max() - base, 1)))
print('entries per eighth of the span:', h.tolist())
