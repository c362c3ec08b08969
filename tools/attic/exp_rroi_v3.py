"""RoI-Align forward A/B: python tools/exp_rroi_v3.py  (OBBOPS_LIB selects the build)
configs[2] (512 rotated RoIs) and the config-4 shape (2000 horizontal RoIs), Morton order by the library's own
order kernel off / on (OBB_TUNE_RROI_ORDER_MIN), through the Python layers (= what bench.py times)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from aidet_amd import _lib, RRoIAlign, RoIAlign
dev = torch.device('cuda', 0)
feat, rois = bench.make_rroi(dev)
layer = RRoIAlign(7, 0.25, sample_num=2)
rng = np.random.default_rng(2)
x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
hr = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)
hrois = torch.from_numpy(hr).to(dev)
hl = RoIAlign(7, 0.25, sample_num=2)
has_order = hasattr(_lib, 'TUNE_RROI_ORDER_MIN')
for order in ((False, True) if has_order else (None,)):
    if order is not None:
        _lib.set_tuning(_lib.TUNE_RROI_ORDER_MIN, 64 if order else (1 << 40))
    a = bench._time_gpu(lambda: layer(feat, rois), 50)
    b = bench._time_gpu(lambda: hl(feat, hrois), 20)
    print(json.dumps({'lib': os.path.basename(_lib.LIB_PATH), 'morton_order': order, 'rroi_512_ms': round(a, 4), 'roi_2000_ms': round(b, 4)}), flush=True)
