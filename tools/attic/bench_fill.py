"""HBM ceilings that bound rbbox_overlaps: fill (write-only) and copy rates at the 16k x 16k output size."""
import torch
n = 16384 * 16384
x = torch.empty(n, dtype=torch.float32, device='cuda')
y = torch.empty(n, dtype=torch.float32, device='cuda')
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, fn, bytes_ in (('zero_ (hipMemset-like fill)', lambda: x.zero_(), 4 * n), ('fill_(1.5)', lambda: x.fill_(1.5), 4 * n),
                         ('copy_ (read + write)', lambda: x.copy_(y), 8 * n), ('mul_ in place (read + write)', lambda: x.mul_(1.0001), 8 * n)):
    ms = t(fn)
    print('%-32s %.3f ms  %.0f GB/s' % (name, ms, bytes_ / ms / 1e6))
