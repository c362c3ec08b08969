"""Per-call time of the HBB RoIAlign drop-in (V1 / V2) next to RRoIAlign at the configs[2] shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import RRoIAlign
from aidet_amd.roi_layers import RoIAlign
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
feat = torch.from_numpy(rng.standard_normal((1, 256, 256, 256)).astype(np.float32)).to(dev)
r = oracle.synth_boxes(512, 1024, 5)[:, :5]
rr = torch.from_numpy(np.concatenate([np.zeros((512, 1), np.float32), r], 1)).to(dev)
hb = torch.from_numpy(np.concatenate([np.zeros((512, 1), np.float32), r[:, :2] - r[:, 2:4] / 2, r[:, :2] + r[:, 2:4] / 2], 1).astype(np.float32)).to(dev)
def t(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
print('RRoIAlign           %.1f us' % t(lambda: RRoIAlign(7, 0.25, sample_num=2)(feat, rr)))
l1 = RoIAlign(7, 0.25, sample_num=2); l2 = RoIAlign(7, 0.25, sample_num=2, aligned=True)
print('RoIAlign V1 (HBB)   %.1f us' % t(lambda: l1(feat, hb)))
print('RoIAlign V2 aligned %.1f us' % t(lambda: l2(feat, hb)))
