"""Stand-alone timing of rbbox_overlaps 16k x 16k for profiling runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rbbox_overlaps
dev = torch.device('cuda:0')
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
canvas = float(sys.argv[2]) if len(sys.argv) > 2 else 1024.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else m
a = torch.from_numpy(oracle.synth_boxes(m, canvas, 1)[:, :5]).to(dev)
b = torch.from_numpy(oracle.synth_boxes(n, canvas, 2)[:, :5]).to(dev)
for _ in range(3): rbbox_overlaps(a, b)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): rbbox_overlaps(a, b)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print('iou %dx%d ms %.4f  -> %.0f GB/s' % (m, n, ms, 4.0 * m * n / ms / 1e6))
