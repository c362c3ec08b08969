"""Anchor grids of the RPN (host side, pure torch).

Counterpart of ``AnchorGenerator`` (mmdet/core/anchor/anchor_generator.py:4-98), pinned against
the reference's output for the five levels of a 1024 x 1024 tile (tests/golden/detector_parts.npz).

An anchor set is the outer sum of two small tables:

    base  (A, 4)   one box per (ratio, scale) combination, centred on ``(size - 1) / 2`` and
                   ROUNDED to integers (anchor_generator.py:29-56)
    shift (H*W, 4) the cell origin ``(x * stride, y * stride)`` twice, x fastest (:66-83)

``all = (shift[:, None] + base[None]).view(-1, 4)``: the A anchors of cell (0, 0) first, then
cell (0, 1), ...  Both tables are built by broadcasting; ``AnchorGenerator`` only keeps the
reference's constructor and method names on top of the two functions.
"""
import torch


def base_anchors(base_size, scales, ratios, scale_major=True, ctr=None):
    """(len(ratios) * len(scales), 4) rounded xyxy boxes.

    ``scale_major=True`` (the default everywhere in the reference) lets the RATIO vary slowest;
    width and height of a combination are ``size * scale / sqrt(ratio)`` and
    ``size * scale * sqrt(ratio)``, evaluated in the reference's operand order so that the
    rounding sees the same floats."""
    scales = torch.as_tensor(scales, dtype=torch.float32)
    root = torch.sqrt(torch.as_tensor(ratios, dtype=torch.float32))
    centre = (0.5 * (base_size - 1), 0.5 * (base_size - 1)) if ctr is None else ctr
    if scale_major:
        w = (base_size * (1 / root)[:, None] * scales[None, :]).reshape(-1)
        h = (base_size * root[:, None] * scales[None, :]).reshape(-1)
    else:
        w = (base_size * scales[:, None] * (1 / root)[None, :]).reshape(-1)
        h = (base_size * scales[:, None] * root[None, :]).reshape(-1)
    half = torch.stack([0.5 * (w - 1), 0.5 * (h - 1)], dim=-1)              # (A, 2)
    c = torch.tensor([centre[0], centre[1]], dtype=torch.float32)
    return torch.cat([c - half, c + half], dim=-1).round()


def grid_anchors(base, featmap_size, stride, device='cuda'):
    """(H * W * A, 4): ``base`` shifted to every cell of an (H, W) map, row-major over cells."""
    base = base.to(device)
    rows, cols = featmap_size
    ys, xs = torch.meshgrid(torch.arange(rows, device=device) * stride, torch.arange(cols, device=device) * stride,
                            indexing='ij')
    origin = torch.stack([xs, ys], dim=-1).reshape(-1, 2).type_as(base)     # x fastest
    return (origin.repeat(1, 2)[:, None, :] + base[None, :, :]).reshape(-1, 4)


class AnchorGenerator(object):
    """The reference's class shape (constructor arguments, ``base_anchors``,
    ``num_base_anchors``, ``grid_anchors``) over the two functions above.

    Example (anchor_generator.py:6-15):
        >>> self = AnchorGenerator(9, [1.], [1.])
        >>> self.grid_anchors((2, 2), device='cpu')
        tensor([[ 0.,  0.,  8.,  8.],
                [16.,  0., 24.,  8.],
                [ 0., 16.,  8., 24.],
                [16., 16., 24., 24.]])
    """

    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size, self.scale_major, self.ctr = base_size, scale_major, ctr
        self.scales, self.ratios = torch.Tensor(scales), torch.Tensor(ratios)
        self.base_anchors = base_anchors(base_size, scales, ratios, scale_major, ctr)

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def grid_anchors(self, featmap_size, stride=16, device='cuda'):
        return grid_anchors(self.base_anchors, featmap_size, stride, device)
