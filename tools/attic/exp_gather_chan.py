"""Gather backward: 16 vs 32 channels per block (OBB_TUNE_RROI_GATHER_CHAN) at configs[2] and the config-4 shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
hr = torch.from_numpy(np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)).to(dev)
st = _lib.current_stream_handle(dev)
for name, r, kind in (('configs[2]', rois, 0), ('config-4 shape', hr, 1)):
    K = r.shape[0]
    gout = torch.randn(K, 256, 7, 7, device=dev)
    g1 = torch.empty_like(feat)
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev)
    def gather():
        _lib.check(L.obb_roi_backward_gather_f32(gout.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, kind, 7, 7, 0.25, 2, 1.0 if kind == 0 else 0.0,
                                                 g1.data_ptr(), ws.data_ptr(), ws.numel(), st))
    outs = {}
    for ch in (16, 32):
        _lib.set_tuning(_lib.TUNE_RROI_GATHER_CHAN, ch)
        t = bench._time_gpu(gather, 20)
        gather(); outs[ch] = g1.clone()
        print('%-16s %d channels per block: %.4f ms' % (name, ch, t), flush=True)
    print('   equal:', bool(torch.equal(outs[16], outs[32])) or float((outs[16] - outs[32]).abs().max()))
