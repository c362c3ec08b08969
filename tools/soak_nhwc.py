"""One-off soak of the channels-last RoI-Align kernels: (a) the suite's fuzz blocks beyond the ones the suite runs; (b) crowded
maps -- 200 .. 1500 RoIs on 16 .. 64 pixel maps, so that tiles list hundreds of RoIs (several batches, several passes of 512
listed RoIs, the stored / re-read path) -- forward and backward bit-equal to the NCHW kernels.
usage: python tools/soak_nhwc.py FIRST LAST"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import oracle
import test_gpu_fuzz
from aidet_amd import RRoIAlign, roi_layers

a, b = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
dev = torch.device('cuda', 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for it in range(a, b):
    test_gpu_fuzz.test_fuzz_channels_last_roi_align(100 + it)
    rng = np.random.default_rng(777000 + it)
    B = int(rng.integers(1, 3)); C = 4 * int(rng.integers(2, 17))
    H = int(rng.integers(16, 65)); W = int(rng.integers(16, 65))
    K = int(rng.integers(200, 1500))
    scale = 0.25
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    r = oracle.synth_boxes(K, max(W, H) / scale, 31 * it + 1)[:, :5]
    r[:, 2:4] = np.clip(r[:, 2:4] * rng.choice([0.2, 0.5, 1.0]), 4.0, 120.0)     # footprints the plan holds
    rois = np.concatenate([rng.integers(0, B, (K, 1)).astype(np.float32), r], 1).astype(np.float32)
    layer = RRoIAlign(7, scale, sample_num=2)
    fa = t(feat).requires_grad_(True)
    fb = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert roi_layers.is_nhwc(fb)
    oa, ob = layer(fa, t(rois)), layer(fb, t(rois))
    assert torch.equal(oa, ob), ('fwd', it)
    g = rng.standard_normal(ob.shape).astype(np.float32)
    oa.backward(t(g)); ob.backward(t(g))
    assert torch.equal(fb.grad.contiguous(), fa.grad), ('bwd', it, B, C, H, W, K)
    if it % 10 == 0:
        print('round', it, 'crowded map', (B, C, H, W, K), 'ok  %.0f s' % (time.time() - t0), flush=True)
print('soak ok: rounds %d..%d' % (a, b - 1))
