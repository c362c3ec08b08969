"""One config-4 tile under rocprofv3 --kernel-trace: which kernels make up the non-conv stages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aidet_amd.models import build_detector, demo_inputs
dev = torch.device('cuda', 0)
model = build_detector(dev, rotated_nms=False)
img, meta = demo_inputs(dev)
for _ in range(3):
    model.simple_test(img, meta)
torch.cuda.synchronize()
for _ in range(10):
    model.simple_test(img, meta)
torch.cuda.synchronize()
