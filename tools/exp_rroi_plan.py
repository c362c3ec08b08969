"""RoI-Align forward with and without the per-RoI plan pass (OBB_TUNE_RROI_PLAN), through the workspace entry point
(what the Python layers call), against the workspace-free entry point; configs[2] and the config-4 shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from aidet_amd import _lib
dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
feat, rois = bench.make_rroi(dev)
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
hr = torch.from_numpy(np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)).to(dev)
for tag, r, kind, ms_ in (('configs[2] 512 rotated', rois, 0, 1.0), ('config-4 shape 2000 horizontal', hr, 1, 0.0)):
    K = r.size(0)
    out = torch.empty((K, 256, 7, 7), device=dev)
    ref = torch.empty_like(out)
    ws = torch.empty(int(L.obb_roi_forward_workspace_bytes(K)), dtype=torch.uint8, device=dev)
    fp = (ctypes.c_void_p * 1)(feat.data_ptr()); hh = (ctypes.c_int * 1)(256); ww = (ctypes.c_int * 1)(256); sc = (ctypes.c_float * 1)(0.25)

    def plain():
        if kind == 0:
            _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 1.0, ref.data_ptr(), st))
        else:
            _lib.check(L.obb_roi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 0, ref.data_ptr(), st))

    def with_ws():
        _lib.check(L.obb_roi_align_forward_ws_f32(fp, hh, ww, sc, 1, 1, 256, r.data_ptr(), None, K, 7, 7, 2, kind, ms_, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    t_plain = bench._time_gpu(plain, 50) * 1e3
    res = {}
    for knob in (0, 1 << 30):          # (the knob is the fewest RoIs that are planned: 0 = always, huge = never)
        _lib.set_tuning(_lib.TUNE_RROI_PLAN, knob)
        out.zero_()
        res[knob] = bench._time_gpu(with_ws, 50) * 1e3
        assert torch.equal(out, ref), (tag, knob)
    _lib.set_tuning(_lib.TUNE_RROI_PLAN, -1)
    print('%-34s plain entry %.1f us | workspace entry: with plan %.1f us, without %.1f us (bit-equal)' % (tag, t_plain, res[0], res[1 << 30]), flush=True)
