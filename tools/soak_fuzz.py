"""One-off extended run of the bounded fuzz sweeps of tests/test_gpu_fuzz.py (more seeds than the
driver-run blocks 0..5).  usage: python tools/soak_fuzz.py FIRST_BLOCK LAST_BLOCK [gather|atomics]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import test_gpu_fuzz as F
from aidet_amd import _lib


def force(route):
    fused = {'fused': 1024, 'direct': 0, 'sorted': 0, 'listed': 0, 'grid': 0, 'auto': -1}[route]
    direct = {'fused': -1, 'direct': 65536, 'sorted': 0, 'listed': 0, 'grid': 0, 'auto': -1}[route]
    inline = {'fused': -1, 'direct': -1, 'sorted': 1 << 40, 'listed': 0, 'grid': 0, 'auto': -1}[route]
    grid = {'fused': -1, 'direct': -1, 'sorted': 0, 'listed': 0, 'grid': 1, 'auto': -1}[route]
    _lib.set_tuning(_lib.TUNE_NMS_FUSED_MAX, fused)
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, direct)
    _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, inline)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, grid)


a, b = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:      # e.g. `gather`: force the RoI-Align backward formulation (roi_layers.BACKWARD)
    from aidet_amd import roi_layers
    roi_layers.BACKWARD = sys.argv[3]
t0 = time.time()
nms = getattr(F.test_fuzz_rnms_all_routes, '__wrapped__', F.test_fuzz_rnms_all_routes)
ops = getattr(F.test_fuzz_overlaps_and_rroi_align, '__wrapped__', F.test_fuzz_overlaps_and_rroi_align)
for blk in range(a, b):
    nms(blk, force)
    ops(blk)
    print('block', blk, 'ok  %.0f s' % (time.time() - t0), flush=True)
force('auto')
print('soak ok: blocks %d..%d' % (a, b - 1))
