"""Round-5 fixtures, produced by RUNNING the reference (nms_cpu.cpp compiled unmodified into oracle/_ref/nms_cpu.so,
called through the reference's own mmdet/ops/nms/nms_wrapper.py:7-60):

  hbb_nms_nonpos.npz   nms(dets, iou_thr <= 0): under `ovr >= threshold` (nms_cpu.cpp:56) every later box of the array
                       falls to the first one -- disjoint boxes included -- unless its ovr is negative or NaN (inverted
                       boxes); float32 and float64 rows, thresholds 0, -0.25, -1.  Scores are distinct: the reference orders equal
                       scores by an unstable torch.sort (nms_cpu.cpp:21), this repo by index.

Runs only in the build container (needs /root/reference).  Re-run:  python tests/golden/make_golden_r5.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, rand_hbb  # noqa: E402


def main():
    nms_wrapper = load_reference()[0]
    rng = np.random.default_rng(505)
    out = {}
    cases = []
    # well-formed boxes, distinct scores
    for tag, n, canvas in (('a', 300, 400), ('b', 2000, 3000)):
        b = rand_hbb(rng, n, canvas, 4, 60)
        s = ((rng.permutation(n) + 0.5) / n)
        cases.append((tag, np.concatenate([b, s[:, None]], 1)))
    # inverted boxes (x2 < x1 - 1 or y2 < y1 - 1: negative widths -> negative areas -> negative or NaN ovr) among them,
    b = rand_hbb(rng, 400, 300, 4, 60, integer=True)
    inv = rng.choice(400, 60, replace=False)
    b[inv[:30], 2] = b[inv[:30], 0] - rng.integers(2, 40, 30)
    b[inv[30:], 3] = b[inv[30:], 1] - rng.integers(2, 40, 30)
    s = ((rng.permutation(400) + 0.5) / 400)      # (distinct: the order of EQUAL scores is torch.sort's, unspecified without stable=True)
    cases.append(('inv', np.concatenate([b, s[:, None]], 1)))
    # degenerate: zero-area boxes (x2 = x1 - 1: width 0).  ovr(zero-area i, zero-area j) = 0 / 0 = NaN: `NaN >= thr` is
    # false, j survives; ovr(zero-area, positive-area) = 0 / A = 0 >= thr: suppressed.  'zero': the best box has a positive
    # area (everything falls to it); 'zerotop': a zero-area box leads (every zero-area box is kept, nothing else)
    b = rand_hbb(rng, 120, 100, 4, 30, integer=True)
    b[::7, 2] = b[::7, 0] - 1
    s = ((rng.permutation(120) + 0.5) / 120)
    top = int(np.argmax(s))
    s[0], s[top] = s[top], s[0]
    assert np.argmax(s) == 0 and b[0, 2] == b[0, 0] - 1
    cases.append(('zerotop', np.concatenate([b, s[:, None]], 1)))
    s2 = s.copy()
    s2[0], s2[1] = 0.001, 2.0
    cases.append(('zero', np.concatenate([b, s2[:, None]], 1)))
    for tag, d in cases:
        for dt, dn in ((np.float32, 'f32'), (np.float64, 'f64')):
            dets = d.astype(dt)
            out['%s_%s_dets' % (tag, dn)] = dets
            for ti, thr in enumerate((0.0, -0.25, -1.0)):
                keep = nms_wrapper.nms(dets, thr)[1]
                out['%s_%s_keep%d' % (tag, dn, ti)] = np.asarray(keep, dtype=np.int64)
    out['thrs'] = np.array([0.0, -0.25, -1.0], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'hbb_nms_nonpos.npz'), **out)
    for k in sorted(out):
        if 'keep' in k:
            print(k, out[k][:12], len(out[k]))


if __name__ == '__main__':
    main()
