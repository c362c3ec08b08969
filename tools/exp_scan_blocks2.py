import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from aidet_amd import _lib, nms_wrapper
dev = torch.device('cuda', 0)
import workloads, numpy as np
for canvas in (20000, 4096):
    b = torch.from_numpy(workloads.boxes(100000, canvas, 3 if canvas == 20000 else 4)).to(dev)
    for sb in (-1, 256, 512, 1024, 2048, 4096, 8192):
        _lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, sb)
        ts = [bench._time_gpu(lambda: nms_wrapper._run_nms(b, _lib.KIND_RBOX, 0.5), 30) for _ in range(3)]
        print('canvas %5d scan blocks %5d: %.4f ms' % (canvas, sb, min(ts)), flush=True)
_lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, -1)
