"""Timing of RRoIAlign backward (float atomics) at the configs[2] shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import RRoIAlign
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
feat = torch.from_numpy(rng.standard_normal((1, 256, 256, 256)).astype(np.float32)).to(dev).requires_grad_(True)
r = oracle.synth_boxes(512, 1024, 5)[:, :5]
rois = torch.from_numpy(np.concatenate([np.zeros((512, 1), np.float32), r], 1)).to(dev)
layer = RRoIAlign(7, 0.25, sample_num=2)
out = layer(feat, rois)
g = torch.ones_like(out)
for _ in range(3):
    feat.grad = None
    out.backward(g, retain_graph=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    feat.grad = None
    out.backward(g, retain_graph=True)
e.record(); torch.cuda.synchronize()
print('rroi backward ms (incl. zero-fill of the 67 MB gradient)', s.elapsed_time(e) / 10)
