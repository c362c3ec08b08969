"""Grid of the sorted route's scan kernel (OBB_TUNE_NMS_SCAN_BLOCKS): C-ABI rnms latency over n and density."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import workloads
from aidet_amd import _lib
from exp_inline_clip import lat_c

for n in (20000, 50000, 100000, 300000, 1000000):
    for per_box, name in ((63.0, 'sparse'), (13.0, 'dense')):
        d = torch.from_numpy(workloads.boxes(n, (n ** 0.5) * per_box, 5)).cuda()
        row = []
        for sb in (1024, 2048, 4096, 8192, 16384):
            _lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, sb)
            row.append('%d: %.1f' % (sb, lat_c(d, 50 if n <= 100000 else 10)[0]))
        _lib.set_tuning(_lib.TUNE_NMS_SCAN_BLOCKS, -1)
        print('%-7s n=%-8d %s us' % (name, n, '   '.join(row)), flush=True)
