"""Channels-last RoI-Align backward against the NCHW gather at configs[2] and the config-4 shape, through the C ABI.
  python3 tools/exp_nhwc_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib

dev = torch.device('cuda', 0)
L = _lib.lib()
st = _lib.current_stream_handle(dev)
for K in (512, 2000):
    feat, rois = bench.make_rroi(dev, K=K)
    go = torch.randn(K, 256, 7, 7, device=dev)
    gm = torch.empty_like(feat)
    gl = torch.empty((1, 256, 256, 256), device=dev, memory_format=torch.channels_last)
    ws = torch.empty(L.obb_roi_backward_workspace_bytes(K), dtype=torch.uint8, device=dev)
    wl = torch.empty(L.obb_roi_backward_nhwc_workspace_bytes(K, 256, 7, 7), dtype=torch.uint8, device=dev)

    def nchw():
        _lib.check(L.obb_roi_backward_gather_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0,
                                                 gm.data_ptr(), ws.data_ptr(), ws.numel(), st))

    def nhwc():
        _lib.check(L.obb_roi_backward_gather_nhwc_f32(go.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), K, 0, 7, 7, 0.25, 2, 1.0,
                                                      gl.data_ptr(), wl.data_ptr(), wl.numel(), st))
    nchw(); nhwc()
    print('K=%d  bit-equal to the NCHW gather: %s' % (K, bool(torch.equal(gl.contiguous(), gm))))
    res = {'nchw gather': [], 'nhwc gather, 4 waves per tile': [], 'nhwc gather, 8 waves per tile': [], 'nhwc gather, 16 waves per tile': []}
    for _ in range(7):
        res['nchw gather'].append(bench._time_gpu(nchw, 20))
        for gw in (4, 8, 16):
            _lib.set_tuning(_lib.TUNE_RROI_NHWC_GATHER_WAVES, gw)
            res['nhwc gather, %d waves per tile' % gw].append(bench._time_gpu(nhwc, 20))
        _lib.set_tuning(_lib.TUNE_RROI_NHWC_GATHER_WAVES, -1)
    for k, v in res.items():
        print('K=%-5d %-32s median %.4f ms  min %.4f ms' % (K, k, float(np.median(v)), min(v)), flush=True)
