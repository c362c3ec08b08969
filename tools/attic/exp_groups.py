"""Merged NMS of many images in ONE grouped call (config 5): boxes of different images share the
coordinate range, so cross-group AABB pairs abound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms

def lat(d, g, reps=10):
    for _ in range(2): rnms(d, 0.3, groups=g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): k = rnms(d, 0.3, groups=g)[1]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, k.numel()

for n, images in ((100000, 1), (100000, 100), (100000, 1000), (1000000, 1000), (1000000, 5000)):
    per = n // images
    d = oracle.synth_boxes(n, 4000.0 * max(1.0, (per / 2000.0) ** 0.5), 3)   # ~2000 boxes per 4000^2 image
    g = (np.arange(n) // per).astype(np.int32) * 15 + np.random.default_rng(1).integers(0, 15, n).astype(np.int32)
    us, k = lat(torch.from_numpy(d).cuda(), torch.from_numpy(g).cuda())
    print('n=%8d images=%5d (x15 classes)  %10.1f us   kept %d' % (n, images, us, k))
