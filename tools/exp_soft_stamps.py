"""Shader-clock stamps of rounds 256..511 of the position-free Soft-NMS kernel (instrumented build -DOBB_SOFT_STAMPS ->
aidet_amd/lib/libobbops_sst.so): where a round goes (n = 2000, 16 waves x 2 boxes per lane)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_sst.so')
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aidet_amd import _lib
from aidet_amd.nms_wrapper import soft_nms
L = _lib.lib()
rng = np.random.default_rng(2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
x1, y1 = rng.uniform(0, 900, n), rng.uniform(0, 900, n)
hb = np.stack([x1, y1, x1 + rng.uniform(8, 200, n), y1 + rng.uniform(8, 200, n), (rng.permutation(n) + .5) / n], 1).astype(np.float32)
hd = torch.from_numpy(hb).cuda()
for _ in range(3):
    soft_nms(hd, 0.5, min_score=0.05)
print('path', L.obb_soft_nms_last_path())
buf = np.zeros(16 * 256 * 8, dtype=np.uint64)
L.obb_debug_soft_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_soft_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(16, 256, 8).astype(np.int64)
names = ['publish done -> barrier passed', '-> the block\'s pick read (two LDS words)', '-> pick corners read (LDS)', '-> decay', '-> next candidate published']
for w in (0, 7, 15):
    r = s[w]
    ok = (r[:, 5] > r[:, 0]) & (r[:, 0] > 0)
    r = r[ok]
    print('wave', w, 'rounds', len(r))
    for i, nm in enumerate(names):
        d = r[:, i + 1] - r[:, i]
        print('   %-50s median %6d  p90 %6d cycles' % (nm, np.median(d), np.percentile(d, 90)))
    full = r[1:, 0] - r[:-1, 0]
    print('   round to round: median %d p90 %d cycles' % (np.median(full), np.percentile(full, 90)))
