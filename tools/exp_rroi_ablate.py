"""Floor derivation of the RoI-Align forward kernel: the same launch through instrumented builds of the
library (aidet_amd/csrc/obb_rroi.hip, OBB_RROI_ABLATE = 1..4: results are wrong on purpose, only the time is read).
Build them first:  for a in 1 2 3 4; do make -C aidet_amd/csrc EXTRA=-DOBB_RROI_ABLATE=$a OUT=../lib/libobbops_abl$a.so; done
(in a copy of csrc/, or `make clean` between them: the objects are shared).  Every build runs in a child process
(OBBOPS_LIB selects the library)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bench
    from aidet_amd import _lib
    dev = torch.device('cuda', 0)
    L = _lib.lib()
    st = _lib.current_stream_handle(dev)
    feat, rois = bench.make_rroi(dev)
    out = torch.empty((512, 256, 7, 7), device=dev)
    res = {}
    res['cfg3_512_rotated'] = bench._time_gpu(lambda: _lib.check(L.obb_rroi_align_forward_f32(
        feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), 512, 7, 7, 0.25, 2, 1.0, out.data_ptr(), st)), 50) * 1e3
    rng = np.random.default_rng(2)
    x1, y1 = rng.uniform(0, 900, 2000), rng.uniform(0, 900, 2000)
    hb = np.stack([np.zeros(2000), x1, y1, x1 + rng.uniform(8, 200, 2000), y1 + rng.uniform(8, 200, 2000)], 1).astype(np.float32)
    hr = torch.from_numpy(hb).to(dev)
    out2 = torch.empty((2000, 256, 7, 7), device=dev)
    res['cfg4_2000_horizontal'] = bench._time_gpu(lambda: _lib.check(L.obb_roi_align_forward_f32(
        feat.data_ptr(), 1, 256, 256, 256, hr.data_ptr(), 2000, 7, 7, 0.25, 2, 0, out2.data_ptr(), st)), 20) * 1e3
    print(json.dumps(res))
    sys.exit(0)

names = {0: 'product build', 1: 'no sample-table arithmetic',
         3: 'one LDS read per output instead of 16 taps', 4: 'no footprint copy (LDS-DMA)', 6: 'no output stores'}
for a in (0, 1, 3, 4, 6):
    lib = os.path.join(ROOT, 'aidet_amd', 'lib', 'libobbops.so' if a == 0 else 'libobbops_abl%d.so' % a)
    if not os.path.exists(lib):
        print(a, 'missing', lib)
        continue
    env = dict(os.environ, OBBOPS_LIB=lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(a, 'failed', r.stderr[-500:])
        continue
    j = json.loads(line[-1])
    print('ablate %d  %-62s configs[2] %.1f us   config-4 shape %.1f us' % (a, names[a], j['cfg3_512_rotated'], j['cfg4_2000_horizontal']), flush=True)
