"""Sorted-route rnms with the candidates clipped in line by the scan (default) vs listed and clipped by
a separate launch (OBB_TUNE_NMS_INLINE_MAX = 0): C-ABI latency over n and scene density, and a
clustered scene (half of the boxes inside 2 % of the canvas)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import workloads
from aidet_amd import _lib

L = _lib.lib()


def lat_c(d, reps):
    n = d.shape[0]
    keep = torch.empty(n, dtype=torch.int64, device='cuda')
    cap = 0
    while True:
        ws = torch.empty(L.obb_nms_workspace_bytes(n, 0, cap), dtype=torch.uint8, device='cuda')
        info = (ctypes.c_int64 * 4)()
        st = torch.cuda.current_stream().cuda_stream
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        if rc == _lib.OBB_EWORKSPACE:
            cap = info[3]
            continue
        assert rc == 0, rc
        break
    def call():
        rc = L.obb_nms_f32(d.data_ptr(), n, 0, 0.5, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), st)
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, info[0], info[1], keep[:info[0]].clone()


def clustered(n, canvas, seed):
    d = workloads.boxes(n, canvas, seed)
    rng = np.random.default_rng(seed + 1)
    m = n // 2
    d[:m, 0] = rng.uniform(0.40, 0.42, m) * canvas * 7 + 0   # a dense strip: 2 % of the width ...
    d[:m, 0] = canvas * 0.4 + rng.uniform(0, 0.14, m) * canvas
    d[:m, 1] = canvas * 0.4 + rng.uniform(0, 0.14, m) * canvas  # ... x 14 % of both sides = 2 % of the area
    return d


if __name__ == '__main__':
    scenes = []
    for n in (20000, 50000, 100000, 300000, 1000000):
        for per_box, name in ((63.0, 'sparse'), (13.0, 'dense'), (4.0, 'very dense')):
            if per_box == 4.0 and n > 100000:
                continue
            scenes.append(('%s n=%d' % (name, n), workloads.boxes(n, (n ** 0.5) * per_box, 5)))
    scenes.append(('clustered n=100000', clustered(100000, 20000.0, 7)))
    scenes.append(('bench_dense n=100000 canvas 4000', workloads.boxes(100000, 4000.0, 5)))
    for name, arr in scenes:
        d = torch.from_numpy(arr).cuda()
        reps = 50 if d.shape[0] <= 100000 else 10
        _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, 0)
        t0, k0, e0, keep0 = lat_c(d, reps)
        _lib.set_tuning(_lib.TUNE_NMS_INLINE_MAX, -1)
        t1, k1, e1, keep1 = lat_c(d, reps)
        assert k0 == k1 and e0 == e1 and torch.equal(keep0, keep1)
        print('%-36s kept %8d edges %9d   listed+clip %9.1f us   in line %9.1f us   (%.2fx)' % (name, k0, e0, t0, t1, t0 / t1), flush=True)
