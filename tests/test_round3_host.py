"""CPU tests of the round-3 host-side additions: the differentiable rotated IoU that backs the gradient of
``rbbox_overlaps`` (second algorithm: candidate vertices + shoelace, vs the oracle's fp64 routine), and the row-block
partition helper."""
import numpy as np
import pytest
import torch

import oracle
from aidet_amd.rbbox_geometry import rbbox_overlaps_torch, rbox_corners


@pytest.mark.parametrize('mode', ['iou', 'iof'])
def test_torch_rotated_iou_equals_the_fp64_oracle(mode):
    a = oracle.synth_boxes(4000, 300, 1)[:, :5]
    b = oracle.synth_boxes(4000, 300, 2)[:, :5]
    got = rbbox_overlaps_torch(torch.from_numpy(a).double(), torch.from_numpy(b).double(), mode, True).numpy()
    ref = oracle.overlaps_f64(a, b, mode=mode, is_aligned=True)
    assert (ref > 0).sum() > 200 and np.abs(got - ref).max() < 1e-10
    got = rbbox_overlaps_torch(torch.from_numpy(a[:150]).double(), torch.from_numpy(b[:220]).double(), mode).numpy()
    assert got.shape == (150, 220) and np.abs(got - oracle.overlaps_f64(a[:150], b[:220], mode=mode)).max() < 1e-10
    # fp32 evaluation stays within the north star's tolerance
    g32 = rbbox_overlaps_torch(torch.from_numpy(a), torch.from_numpy(b), mode, True).numpy()
    assert np.abs(g32 - ref).max() < 1e-4


def test_torch_rotated_iou_analytic_cases():
    t = lambda *r: torch.tensor(r, dtype=torch.float64)
    one = t([0., 0., 1., 1., 0.])
    f = lambda x, y: float(rbbox_overlaps_torch(x, y, 'iou', True)[0])
    assert abs(f(one, one) - 1.0) < 1e-12                                   # identical
    assert abs(f(one, t([0., 0., 1., 1., np.pi / 4])) - 2 ** 0.5 / 2) < 1e-12   # regular octagon
    assert f(one, t([5., 0., 1., 1., 0.3])) == 0.0                           # disjoint
    assert abs(f(t([0., 0., 4., 4., 0.2]), t([0., 0., 2., 2., 0.2])) - 0.25) < 1e-12   # containment
    assert f(one, t([0., 0., 0., 3., 0.])) == 0.0                            # zero area: 0, never NaN
    assert rbox_corners(t([10., 20., 4., 2., 0.])).tolist() == [[[8., 19.], [12., 19.], [12., 21.], [8., 21.]]]


def test_torch_rotated_iou_gradient_matches_finite_differences():
    a = torch.from_numpy(oracle.synth_boxes(48, 200, 3)[:, :5]).double()
    b = (a + torch.tensor([3., 2., 1., -1., 0.2], dtype=torch.float64)).clone()
    a.requires_grad_(True); b.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda u, v: rbbox_overlaps_torch(u, v, 'iou', True), (a, b), eps=1e-6, atol=1e-5, rtol=1e-4)
    assert torch.autograd.gradcheck(lambda u, v: rbbox_overlaps_torch(u, v, 'iof', True), (a, b), eps=1e-6, atol=1e-5, rtol=1e-4)
    # an IoU loss decreases under gradient descent on the predicted box
    pred = torch.tensor([[12., 9., 30., 12., -0.2]], dtype=torch.float64, requires_grad=True)
    gt = torch.tensor([[10., 10., 28., 14., -0.5]], dtype=torch.float64)
    first = None
    for _ in range(60):
        loss = 1 - rbbox_overlaps_torch(pred, gt, 'iou', True).sum()
        first = float(loss) if first is None else first
        g, = torch.autograd.grad(loss, pred)
        pred = (pred - 0.5 * g).detach().requires_grad_(True)
    assert float(loss) < 0.5 * first
