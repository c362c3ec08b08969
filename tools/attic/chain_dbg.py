import sys
sys.path.insert(0, '.')
import numpy as np, torch
from aidet_amd import _lib, rnms
n = 2000
d = np.zeros((n, 6), np.float32)
d[:, 0] = np.arange(n) * 4.0; d[:, 1] = 5; d[:, 2] = 10; d[:, 3] = 10; d[:, 5] = 1.0 - np.arange(n) / n
t = torch.from_numpy(d).cuda()
for mode in (2, 0, 3, 1):
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)
    _, inds = rnms(t, 0.4)
    print('mode', mode, 'kept', inds.numel(), 'last', int(inds[-1]), 'route', _lib.lib().obb_nms_last_route())
import time
for mode in (0, 1):
    _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, inds = rnms(t, 0.4)
    torch.cuda.synchronize()
    print('mode', mode, 'kept', inds.numel(), 'seconds %.4f' % (time.perf_counter() - t0))
