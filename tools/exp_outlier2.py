import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms
os.environ['OBB_NMS_DEBUG'] = '1'
n, canvas = 100000, 20000
base = oracle.synth_boxes(n, canvas, 3)
for far, cnt in ((0, 0), (1e6, 1), (1e6, n // 100)):
    d = base.copy()
    if cnt:
        idx = np.arange(cnt) * (n // cnt)
        d[idx, 0] += far
        d[idx, 1] -= far * 0.5
    rnms(torch.from_numpy(d).cuda(), 0.5)
