"""Shader-clock stamps of every wave of the grid route's probe kernel (instrumented build -DOBB_GPROBE_STAMPS of
aidet_amd/csrc/obb_nms.hip -> aidet_amd/lib/libobbops_gpst.so; tools/build_ablations.sh gpst): where a wave's life goes
at the headline workload (100k theta-OBBs, canvas 20000)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_gpst.so')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import workloads
from aidet_amd import _lib, nms_wrapper
dev = torch.device('cuda', 0)
L = _lib.lib()
scene = sys.argv[1] if len(sys.argv) > 1 else '100000'
if scene == 'merge':      # the merged NMS of a config-5 pass: 1.25 M rows, group = image x class, per-class thresholds
    from aidet_amd import dota_io, dota_merge
    sc = workloads.tiled_scene(images=16)
    args = [torch.from_numpy(sc[k]).to(dev) for k in ('dets', 'label', 'patch')]
    rows, counts = dota_merge.tiled_local_phase(*args, torch.from_numpy(sc['patch_xy']).to(dev), torch.from_numpy(sc['patch_image']).to(dev), 15, 0.5, 1)
    K = counts.tolist()[2]
    r = rows[:K].contiguous()
    d = r[:, :6].contiguous()
    g = (r[:, 7].long() * 15 + r[:, 6].long()).int()
    t = dota_merge._class_thr_tensor(dota_io.class_thresholds(), dev)[r[:, 6].long()].contiguous()
    for _ in range(5):
        keep = nms_wrapper.rnms(d, 0.3, groups=g, thrs=t)[1]
else:
    n = int(scene)
    d = torch.from_numpy(workloads.boxes(n, 20000.0, 3)).to(dev)
    for _ in range(10):
        keep = nms_wrapper.rnms(d, 0.5)[1]
torch.cuda.synchronize()
W = 32768
buf = np.zeros(W * 10, dtype=np.uint64)
L.obb_debug_gprobe_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_gprobe_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(W, 10).astype(np.int64)
live = s[:, 7] > 0
print('route', _lib.lib().obb_nms_last_route(), 'kept', len(keep), 'waves with stamps', int((s[:, 0] > 0).sum()), 'that reached the end', int(live.sum()))
w = s[live]
span = (w[:, 9].max() - s[s[:, 8] > 0][:, 8].min()) / 100.0
print('first wave entry -> last wave exit: %.1f us (100 MHz wall clock)' % span)
print('wave entry after the first entry (us): median %.1f  p90 %.1f  max %.1f' % tuple(np.percentile((w[:, 8] - s[s[:, 8] > 0][:, 8].min()) / 100.0, [50, 90, 100])))
print('wave life (us): median %.1f  p90 %.1f  max %.1f' % tuple(np.percentile((w[:, 9] - w[:, 8]) / 100.0, [50, 90, 100])))
if scene == 'merge':
    life = (w[:, 7] - w[:, 0])
    print('waves %d; wave life (clocks): median %d p90 %d max %d' % (len(w), np.median(life), np.percentile(life, 90), life.max()))
    print('  in-loop flushes: median %d clocks = %.0f %% of the life, %d flushes; overflow pass: median %d clocks (%.0f %%), p90 %d' % (
        np.median(w[:, 4]), 100.0 * w[:, 4].sum() / life.sum(), np.median(w[:, 6]), np.median(w[:, 5]), 100.0 * w[:, 5].sum() / life.sum(), np.percentile(w[:, 5], 90)))
    print('  entry -> first chunk %.0f %%; the rest (chunk loads, pair tests, last flush)' % (100.0 * (w[:, 1] - w[:, 0]).sum() / life.sum()))
    sys.exit(0)
names = ['entry -> frame record, fail word', '-> cell counts + entries', '-> pair tests of the cells', '-> records loaded, pruned', '-> clipped', '-> edges pushed', '-> exit']
full = w[(w[:, 4] > 0) & (w[:, 5] > 0) & (w[:, 6] > 0)]
print('%d waves with a flush that clipped' % len(full))
tot = full[:, 7] - full[:, 0]
for i, nm in enumerate(names):
    dd = full[:, i + 1] - full[:, i]
    print('%-40s median %6d  mean %6d  p90 %6d cycles (%.0f %%)' % (nm, np.median(dd), dd.mean(), np.percentile(dd, 90), 100.0 * dd.sum() / tot.sum()))
print('wave life in shader cycles: median %d p90 %d max %d' % tuple(np.percentile(tot, [50, 90, 100])))
