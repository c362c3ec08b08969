import os, sys, torch
sys.path.insert(0, ".")
import workloads
from aidet_amd import _lib
from aidet_amd.nms_wrapper import _run_nms
_lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
d = torch.from_numpy(workloads.boxes(100000, 20000.0, 3)).cuda()
for _ in range(30): _run_nms(d, _lib.KIND_RBOX, 0.5)
torch.cuda.synchronize()
