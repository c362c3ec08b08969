"""rnms at 100k boxes (sparse / dense) over the scan grid (OBB_TUNE_NMS_SCAN_BLOCKS), and the in-line route at 20k / 50k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, workloads
from aidet_amd import _lib, nms_wrapper
dev = torch.device('cuda', 0)
for n, canvas, seed in ((100000, 20000, 3), (100000, 4096, 4), (1000000, 63000, 3)):
    b = torch.from_numpy(workloads.boxes(n, canvas, seed)).to(dev)
    for sb in (-1, 512, 1024, 2048, 4096, 8192):
        _lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, sb)
        ts = [bench._time_gpu(lambda: nms_wrapper._run_nms(b, _lib.KIND_RBOX, 0.5), 30) for _ in range(3)]
        print('n %7d canvas %5d scan blocks %5d: %.4f ms' % (n, canvas, sb, min(ts)), flush=True)
_lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, -1)
for n, canvas in ((20000, 9000), (50000, 14000), (20000, 2000)):
    b = torch.from_numpy(workloads.boxes(n, canvas, 3)).to(dev)
    ts = [bench._time_gpu(lambda: nms_wrapper._run_nms(b, _lib.KIND_RBOX, 0.5), 50) for _ in range(3)]
    print('n %7d canvas %5d (default route): %.4f ms' % (n, canvas, min(ts)), flush=True)
