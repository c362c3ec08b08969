"""Stamps of the first two launches of the NMS's grid route at the headline workload (instrumented build
-DOBB_FRAME_STAMPS -> aidet_amd/lib/libobbops_frst.so, tools/build_ablations.sh): the phases of block 0 of
nms_frame_kernel (shader clock) and begin / end of every block of nms_frame_kernel and nms_ginsert_kernel (wall clock,
100 MHz)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.environ.get('OBB_FRST_LIB') or os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_frst.so')      # (OBB_FRST_LIB: another instrumented build)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import workloads
from aidet_amd import _lib, nms_wrapper
dev = torch.device('cuda', 0)
L = _lib.lib()
stateless = 'stateless' in sys.argv[1:]                      # the four-launch call (OBB_NMS_RETAIN ignored): the frame kernel's phases
if stateless:
    sys.argv.remove('stateless')
grouped = len(sys.argv) > 1 and sys.argv[1] == 'merge'      # the merged NMS of a config-5 pass (group = image x class, per-class thresholds)
if grouped:
    from aidet_amd import dota_io, dota_merge
    sc = workloads.tiled_scene(images=16)
    args = [torch.from_numpy(sc[k]).to(dev) for k in ('dets', 'label', 'patch')]
    rows, counts = dota_merge.tiled_local_phase(*args, torch.from_numpy(sc['patch_xy']).to(dev), torch.from_numpy(sc['patch_image']).to(dev), 15, 0.5, 1)
    K = counts.tolist()[2]
    r = rows[:K].contiguous()
    d = r[:, :6].contiguous()
    g = (r[:, 7].long() * 15 + r[:, 6].long()).int()
    t = dota_merge._class_thr_tensor(dota_io.class_thresholds(), dev)[r[:, 6].long()].contiguous()
    n = K
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    d = torch.from_numpy(workloads.boxes(n, 20000.0 * (n / 100000.0) ** 0.5, 3)).to(dev)
_lib.set_tuning(_lib.TUNE_NMS_RETAIN, 0 if stateless else 1)
L.obb_debug_frame_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ['entry -> own share zeroed', '-> samples in LDS (global round trip)', '-> one-pass statistics', '-> trimming rounds (outliers only)',
         '-> trimmed box, group hashes', '-> frame written (thread 0)']
acc = []
for rep in range(12):
    keep = nms_wrapper.rnms(d, 0.3, groups=g, thrs=t)[1] if grouped else nms_wrapper.rnms(d, 0.5)[1]
    torch.cuda.synchronize()
    buf = np.zeros(16 + 4 * 4096, dtype=np.uint64)
    assert L.obb_debug_frame_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.astype(np.int64)
    w = s[16:].reshape(2, 4096, 2)
    fr, ins = w[0][w[0][:, 0] > 0], w[1][w[1][:, 0] > 0]
    t0 = fr[:, 0].min()
    row = [s[i + 1] - s[i] for i in range(6)] + [
        (fr[0, 1] - fr[0, 0]) / 100.0, (fr[:, 1].max() - t0) / 100.0, (ins[:, 0].min() - t0) / 100.0, (ins[:, 0].max() - t0) / 100.0,
        (np.median(ins[:, 1] - ins[:, 0])) / 100.0, (ins[:, 1].max() - t0) / 100.0, len(fr), len(ins)]
    if rep >= 2:
        acc.append(row)
a = np.median(np.array(acc, dtype=np.float64), axis=0)
print('route', L.obb_nms_last_route(), 'kept', len(keep), 'n', n, 'retained', L.obb_nms_last_retained())
if L.obb_nms_last_retained() == 2:
    # a three-launch call: no frame kernel (its stamps are the seeding call's); the insert launch is the call's first
    print('insert kernel of a three-launch call (the first launch): blocks begin within %.1f us, median block life %.1f us, last ends %.1f us after the first began (%d blocks incl. the frame block)'
          % (a[9] - a[8], a[10], a[11] - a[8], a[13]))
    sys.exit(0)
print('nms_frame_kernel, block 0, thread 0 (shader clocks, median of 10 calls):')
for i, nm in enumerate(names):
    print('  %-44s %7d' % (nm, a[i]))
print('block 0 of the frame kernel alive %.1f us; last frame block ends at %.1f us after the first began (%d blocks)' % (a[6], a[7], a[12]))
print('insert kernel: first block begins at %.1f us, last begins at %.1f, median block life %.1f us, last ends at %.1f us (%d blocks)' % (a[8], a[9], a[10], a[11], a[13]))
