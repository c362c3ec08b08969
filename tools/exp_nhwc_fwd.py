"""Channels-last RoI-Align forward against the NCHW kernel at configs[2] (512 rotated RoIs x 256 ch, 256^2 map, 7 x 7)
and at the config-4 shape (2000 horizontal RoIs over a 4-level pyramid):  python3 tools/exp_nhwc_fwd.py
Interleaved rounds in one process (guide 5.4 rule 24); prints median / min ms per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib, RRoIAlign, roi_layers

dev = torch.device('cuda', 0)


def timed(fn, iters=20):
    return bench._time_gpu(fn, iters)


def run(label, variants, rounds=7):
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            res[k].append(timed(fn))
    for k, v in res.items():
        print('%-28s %-26s median %.4f ms  min %.4f ms' % (label, k, float(np.median(v)), min(v)), flush=True)


def main():
    feat, rois = bench.make_rroi(dev)
    fl = feat.contiguous(memory_format=torch.channels_last)
    layer = RRoIAlign(7, 0.25, sample_num=2)
    ref = layer(feat, rois)
    assert torch.equal(layer(fl, rois), ref)
    variants = {'nchw': lambda: layer(feat, rois)}
    L = _lib.lib()
    st = _lib.current_stream_handle(dev)
    out_f = torch.empty((512, 256, 7, 7), device=dev)
    for nq in (64, 32, 16, 8):
        for nw in (4, 8):
            def f(nq=nq, nw=nw):     # through the C ABI: the Python layer's per-call overhead is as long as this kernel
                _lib.set_tuning(_lib.TUNE_RROI_NHWC_NQ, nq)
                _lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, nw)
                _lib.check(L.obb_rroi_align_forward_nhwc_f32(fl.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0,
                                                             out_f.data_ptr(), st))
            f()
            assert torch.equal(out_f, ref)
            variants['nhwc nq=%d waves=%d (C ABI)' % (nq, nw)] = f
    run('configs[2] 512 RoIs', variants)
    _lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, -1)
    # 2000 RoIs on one 256^2 map (the config-4 shape's first level, covered 22x)
    feat2, rois2 = bench.make_rroi(dev, K=2000)
    fl2 = feat2.contiguous(memory_format=torch.channels_last)
    assert torch.equal(layer(fl2, rois2), layer(feat2, rois2))
    variants = {'nchw': lambda: layer(feat2, rois2)}
    for nq in (64, 32, 16):
        for nw in (4, 8):
            def f(nq=nq, nw=nw):
                _lib.set_tuning(_lib.TUNE_RROI_NHWC_NQ, nq)
                _lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, nw)
                return layer(fl2, rois2)
            variants['nhwc nq=%d waves=%d' % (nq, nw)] = f
    run('2000 RoIs, 256^2 map', variants)


if __name__ == '__main__':
    main()
