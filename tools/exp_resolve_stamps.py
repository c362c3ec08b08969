"""Per-sweep stamps of the resolve kernel (instrumented build OBB_RESOLVE_STAMPS -> aidet_amd/lib/libobbops_rsst.so) on the
merged NMS of a config-5 pass: live edges and wall clock of every sweep of every shard.
  python3 tools/exp_resolve_stamps.py [images] [tail]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['OBBOPS_LIB'] = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops_rsst.so')
sys.path.insert(0, ROOT)
import numpy as np, torch
import workloads
from aidet_amd import _lib, dota_io, dota_merge
from aidet_amd.nms_wrapper import _run_nms
dev = torch.device('cuda', 0)
if len(sys.argv) > 1 and sys.argv[1] == 'headline':      # the headline call (100 000 boxes, ~270 edges): entry and end-of-sweeps of the 64 shards
    d = torch.from_numpy(workloads.boxes(100000, 20000.0, 3)).to(dev)
    L = _lib.lib()
    for _ in range(6):
        keep, info = _run_nms(d, _lib.KIND_RBOX, 0.5)
    torch.cuda.synchronize()
    buf = np.zeros(64 * 64 * 2, dtype=np.uint64)
    L.obb_debug_resolve_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.obb_debug_resolve_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(64, 64, 2).astype(np.int64)
    t_in, t_out = s[:, 0, 0], s[:, 63, 0]
    base = t_in.min()
    print('headline: kept %d edges %d rounds %d retained %d' % (keep.numel(), info[0], info[1], L.obb_nms_last_retained()))
    print('shards enter within %.2f us; the sweeps of a shard end %.2f .. %.2f us after the first entry (median %.2f): what follows in the launch is the '
          'arrival, the slices\' compaction and the record' % ((t_in.max() - base) / 100.0, (t_out.min() - base) / 100.0, (t_out.max() - base) / 100.0,
                                                             float(np.median(t_out - base)) / 100.0))
    sys.exit(0)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, int(sys.argv[2]))
scene = workloads.tiled_scene(images=images)
C = scene['num_classes']
args = [torch.from_numpy(scene[k]).to(dev) for k in ('dets', 'label', 'patch')]
rows, counts = dota_merge.tiled_local_phase(*args, torch.from_numpy(scene['patch_xy']).to(dev), torch.from_numpy(scene['patch_image']).to(dev), C, 0.5, 1)
K = counts.tolist()[2]
r = rows[:K]
d = r[:, :6].contiguous()
grp = (r[:, 7].to(torch.int64) * C + r[:, 6].to(torch.int64)).to(torch.int32)
thr = torch.tensor(dota_io.class_thresholds(), device=dev, dtype=torch.float32)[r[:, 6].to(torch.int64)]
L = _lib.lib()
for _ in range(4):
    keep, info = _run_nms(d, _lib.KIND_RBOX, 0.3, grp, thr)
torch.cuda.synchronize()
buf = np.zeros(64 * 64 * 2, dtype=np.uint64)
L.obb_debug_resolve_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.obb_debug_resolve_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(64, 64, 2).astype(np.int64)
t_in, live0 = s[:, 0, 0], s[:, 0, 1]
t_out, rounds = s[:, 63, 0], s[:, 63, 1]
base = t_in.min()
print('rows %d kept %d edges %d rounds %d' % (K, keep.numel(), info[0], info[1]))
print('shards: edges min %d median %d max %d; entry %d..%d, exit %d..%d (10 ns ticks); sweeps %d..%d' %
      (live0.min(), np.median(live0), live0.max(), t_in.min() - base, t_in.max() - base, t_out.min() - base, t_out.max() - base, rounds.min(), rounds.max()))
slow = int(np.argmax(t_out))
for sh in (slow, int(np.argmin(t_out))):
    print('shard %d (%s to leave): ' % (sh, 'last' if sh == slow else 'first'), end='')
    prev = t_in[sh]
    for k in range(1, 62):
        lv, tk = s[sh, k]
        if lv == 0 and tk == 0:
            break
        print('[sweep %d: %d live, +%d] ' % (k, lv, tk - prev), end='')
        prev = tk
    print('exit +%d' % (t_out[sh] - prev))
