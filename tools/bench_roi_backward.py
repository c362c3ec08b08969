"""RoI-Align backward: the gather formulation (obb_roi_backward_gather_f32: plan per RoI + one block per
map tile, no atomics, overwrites the map) vs the scatter kernels (atomics into a zero-filled map), at
configs[2] (512 rotated RoIs) and at the config-4 shape (2000 horizontal RoIs of 8-200 px on the 256^2 level)."""
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
for name, r, kind in (('configs[2]: 512 rotated RoIs', rois, 0), ('config-4 shape: 2000 horizontal RoIs (V1)', hr, 1)):
    K = r.shape[0]
    gout = torch.randn(K, 256, 7, 7, device=dev)
    g1 = torch.empty_like(feat); g2 = torch.zeros_like(feat)
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev)
    def gather():
        _lib.check(L.obb_roi_backward_gather_f32(gout.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, kind, 7, 7, 0.25, 2, 1.0 if kind == 0 else 0.0,
                                                 g1.data_ptr(), ws.data_ptr(), ws.numel(), st))
    def scatter():
        g2.zero_()
        if kind == 0:
            _lib.check(L.obb_rroi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 1.0, g2.data_ptr(), st))
        else:
            _lib.check(L.obb_roi_align_backward_f32(gout.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 0, g2.data_ptr(), st))
    tg, ts = bench._time_gpu(gather, 20), bench._time_gpu(scatter, 20)
    gather(); a = g1.clone(); gather(); scatter()
    # RoIs whose footprint exceeds the plan (2047 pixels) are added by the atomics kernel: only they make two runs differ
    rn = r.cpu().numpy()
    if kind == 1:
        fw = np.floor(rn[:, 3] * 0.25) - np.floor(rn[:, 1] * 0.25) + 2; fh = np.floor(rn[:, 4] * 0.25) - np.floor(rn[:, 2] * 0.25) + 2
        big = int(((fw * fh) > 2047).sum())
    else:
        big = 0
    print('%-44s gather %.4f ms   scatter (incl. zero fill) %.4f ms   max |diff| %.2e of max|g| %.2f   gather twice bit-equal: %s (about %d RoIs beyond the plan -> atomics)'
          % (name, tg, ts, float((g1 - g2).abs().max()), float(g2.abs().max()), bool(torch.equal(a, g1)), big), flush=True)
