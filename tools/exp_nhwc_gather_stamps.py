"""Per-tile stamps of the channels-last gather backward (instrumented build OBB_NHWC_STAMPS -> aidet_amd/lib/libobbops_nhst.so)
at configs[2]: is the launch as long as its heaviest tile, or as the sum of its tiles?
  python3 tools/exp_nhwc_gather_stamps.py [K]"""
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
feat, rois = bench.make_rroi(dev)
if K != 512:
    import workloads
    rois = torch.from_numpy(np.concatenate([np.zeros((K, 1), np.float32), workloads.boxes(K, 1024, 9)[:, :5]], 1)).to(dev)
go = torch.randn(K, 256, 7, 7, device=dev)
gm = torch.empty_like(feat)
ws = torch.empty(L.obb_roi_backward_nhwc_workspace_bytes(K, 256, 7, 7), dtype=torch.uint8, device=dev)
run = lambda: _lib.check(L.obb_roi_backward_gather_nhwc_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0,
                                                               gm.data_ptr(), ws.data_ptr(), ws.numel(), st))
for _ in range(20):
    run()
torch.cuda.synchronize()
ms = bench._time_gpu(run, 50)
run(); torch.cuda.synchronize()
buf = np.zeros(12 * 4096, dtype=np.uint64)
L.obb_debug_nhwc_gather_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_nhwc_gather_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(4096, 12).astype(np.int64)[:1024]
print('K = %d: whole call %.1f us (stamped build); %d tile blocks' % (K, ms * 1e3, len(s)))
ent, nroi, nbat = s[:, 9], s[:, 8], s[:, 10]
print('entries per tile: total %d, mean %.0f, median %d, p90 %d, p99 %d, max %d; RoIs per tile mean %.1f max %d; batches max %d'
      % (ent.sum(), ent.mean(), np.median(ent), np.percentile(ent, 90), np.percentile(ent, 99), ent.max(), nroi.mean(), nroi.max(), nbat.max()))
t0, t1 = s[:, 6], s[:, 7]
base = t0.min()
life = t1 - t0
print('real time (10 ns ticks): last entry %d, first exit %d, median exit %d, p90 exit %d, last exit %d' %
      (t0.max() - base, t1.min() - base, np.median(t1 - base), np.percentile(t1 - base, 90), t1.max() - base))
has = nbat > 0
names = ['entry -> list of RoIs', '-> bounds + scan (batch 1)', '-> fill (batch 1)', '-> stream (batch 1)', '-> later batches, exit']
tot = s[has, 5] - s[has, 0]
for i, nm in enumerate(names):
    d = s[has, i + 1] - s[has, i]
    print('%-30s median %6d  mean %6d  p90 %6d  max %6d cycles (%.0f %% of the blocks)' % (nm, np.median(d), d.mean(), np.percentile(d, 90), d.max(), 100.0 * d.sum() / tot.sum()))
print('block life (ticks) by entries: ', end='')
for lo, hi in ((0, 1), (1, 100), (100, 200), (200, 400), (400, 800), (800, 100000)):
    m = (ent >= lo) & (ent < hi)
    if m.any():
        print('[%d,%d): n=%d median %d max %d | ' % (lo, hi, m.sum(), np.median(life[m]), life[m].max()), end='')
print()
o = np.argsort(-life)[:8]
print('slowest tiles: ' + ', '.join('tile %d: %d entries, %d RoIs, %d batches, life %d ticks, stream %d cycles' %
                                   (i, ent[i], nroi[i], nbat[i], life[i], s[i, 4] - s[i, 3]) for i in o))
