import time, numpy as np, torch, sys
sys.path.insert(0,'.')
import oracle
from aidet_amd.nms_wrapper import soft_nms
rng=np.random.default_rng(2)
for n in (500, 2000, 2048, 5000):
    x1,y1=rng.uniform(0,900,n),rng.uniform(0,900,n)
    hb=np.stack([x1,y1,x1+rng.uniform(8,200,n),y1+rng.uniform(8,200,n),(rng.permutation(n)+.5)/n],1).astype(np.float32)
    hd=torch.from_numpy(hb).cuda()
    for method in ('linear', 'gaussian'):
        ts=[]
        for _ in range(8):
            t0=time.perf_counter(); sd,si=soft_nms(hd,0.5,method=method,min_score=0.05); ts.append(time.perf_counter()-t0)
        tcs=[]
        for _ in range(3):
            t0=time.perf_counter(); rd,ri=oracle.soft_nms(hb,0.5,method,0.5,0.05); tcs.append(time.perf_counter()-t0)
        print(n, method, 'gpu ms %.3f'%(np.median(ts[2:])*1e3), 'cpu ms %.3f'%(min(tcs)*1e3), 'kept', len(ri),
              np.array_equal(si.cpu().numpy(),ri) and np.array_equal(sd.cpu().numpy(),rd))
