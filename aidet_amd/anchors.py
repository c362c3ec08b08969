"""Anchor grids of the RPN (host side, pure torch).

Counterpart of ``AnchorGenerator`` (mmdet/core/anchor/anchor_generator.py:4-98): base anchors
with the ``(w - 1) / 2`` centre and ``.round()`` (:29-56), then one shift per feature-map
cell, x fastest, ``A`` anchors per cell (:66-83).  Pinned against the reference's output for
the five levels of a 1024 x 1024 tile (tests/golden/detector_parts.npz).
"""
import torch


class AnchorGenerator(object):
    """
    Example (anchor_generator.py:6-15):
        >>> self = AnchorGenerator(9, [1.], [1.])
        >>> self.grid_anchors((2, 2), device='cpu')
        tensor([[ 0.,  0.,  8.,  8.],
                [16.,  0., 24.,  8.],
                [ 0., 16.,  8., 24.],
                [16., 16., 24., 24.]])
    """

    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.Tensor(scales)
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        size = self.base_size
        cx, cy = (0.5 * (size - 1), 0.5 * (size - 1)) if self.ctr is None else self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:   # ratio varies slowest
            ws = (size * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (size * h_ratios[:, None] * self.scales[None, :]).view(-1)
        else:
            ws = (size * self.scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (size * self.scales[:, None] * h_ratios[None, :]).view(-1)
        half_w, half_h = 0.5 * (ws - 1), 0.5 * (hs - 1)
        return torch.stack([cx - half_w, cy - half_h, cx + half_w, cy + half_h], dim=-1).round()

    def grid_anchors(self, featmap_size, stride=16, device='cuda'):
        """(H * W * A, 4): the A base anchors shifted to every cell, row-major over cells."""
        base = self.base_anchors.to(device)
        feat_h, feat_w = featmap_size
        sx = (torch.arange(0, feat_w, device=device) * stride).type_as(base)
        sy = (torch.arange(0, feat_h, device=device) * stride).type_as(base)
        xx = sx.repeat(feat_h)
        yy = sy.repeat_interleave(feat_w)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 4)
