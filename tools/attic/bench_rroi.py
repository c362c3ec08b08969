"""Stand-alone timing of RRoIAlign (BASELINE configs[2]) for profiling runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import RRoIAlign
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
feat = torch.from_numpy(rng.standard_normal((1, 256, 256, 256)).astype(np.float32)).to(dev)
r = oracle.synth_boxes(512, 1024, 5)[:, :5]
rois = torch.from_numpy(np.concatenate([np.zeros((512, 1), np.float32), r], 1)).to(dev)
layer = RRoIAlign(7, 0.25, sample_num=2)
for _ in range(3): layer(feat, rois)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): layer(feat, rois)
e.record(); torch.cuda.synchronize()
print('rroi ms', s.elapsed_time(e) / 20)
