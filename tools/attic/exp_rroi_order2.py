"""RoI-Align forward at the config-4 shape (2000 horizontal RoIs of 8-200 px, 256^2 level) and at configs[2]:
RoIs pre-permuted on the host into (a) Morton order of their centres, (b) size classes (largest first) then Morton,
(c) largest first -- which order should rroi_order_kernel produce?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib
L = _lib.lib()
dev = torch.device('cuda', 0)
feat, rr = bench.make_rroi(dev)
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
hr = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)
st = _lib.current_stream_handle(dev)


def spread(v):
    v = (v | (v << 4)) & 0x0F0F; v = (v | (v << 2)) & 0x3333; v = (v | (v << 1)) & 0x5555
    return v


def run(name, rois_np, kind):
    K = rois_np.shape[0]
    if kind == 1:
        cx, cy = (rois_np[:, 1] + rois_np[:, 3]) / 2, (rois_np[:, 2] + rois_np[:, 4]) / 2
        area = (rois_np[:, 3] - rois_np[:, 1]) * (rois_np[:, 4] - rois_np[:, 2])
    else:
        cx, cy, area = rois_np[:, 1], rois_np[:, 2], rois_np[:, 3] * rois_np[:, 4]
    mort = spread((cx / 32).astype(np.int64).clip(0, 31)) | (spread((cy / 32).astype(np.int64).clip(0, 31)) << 1)
    cls = np.digitize(area, np.quantile(area, [0.5, 0.8, 0.95]))          # 0 small .. 3 largest
    orders = {'given': np.arange(K), 'morton': np.argsort(mort, kind='stable'), 'size class, then morton': np.lexsort((mort, -cls)),
              'largest first': np.argsort(-area, kind='stable')}
    out = torch.empty(K, 256, 7, 7, device=dev)
    for oname, o in orders.items():
        r = torch.from_numpy(rois_np[o]).to(dev).contiguous()
        def fwd():
            if kind == 0:
                _lib.check(L.obb_rroi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st))
            else:
                _lib.check(L.obb_roi_align_forward_f32(feat.data_ptr(), 1, 256, 256, 256, r.data_ptr(), K, 7, 7, 0.25, 2, 0, out.data_ptr(), st))
        print('%-12s %-26s %.4f ms' % (name, oname, min(bench._time_gpu(fwd, 30) for _ in range(3))), flush=True)


run('configs[2]', rr.cpu().numpy(), 0)
run('config-4', hr, 1)
