"""RoI layers built on the rotated RoI-Align kernel.

``RoIAlign``: drop-in for the reference's horizontal op
(mmdet/ops/roi_align/roi_align.py:79-152; its CUDA kernels need THC headers and
cannot be built on ROCm): V1 (``aligned=False`` in the reference means the LEGACY
kernel, roi_align.py:28-36 -> roi_align_kernel.cu:64-117) and V2
(``aligned=True``, roi_align_kernel_v2.cu:62-128) are both served by
``obb_rroi_align_*`` at theta = 0:

  V1:  start = x1*s, end = (x2+1)*s, size clamped at 0
       == rotated RoI  centre ((x1+x2+1)/2, (y1+y2+1)/2), size (x2-x1+1, y2-y1+1), min_size 0
  V2 aligned:  start = x1*s - 0.5, size (x2-x1)*s unclamped
       == rotated RoI  centre ((x1+x2)/2 - 0.5/s, ...), size (x2-x1, y2-y1), min_size 0

``RotatedSingleRoIExtractor``: the multi-level extractor of
mmdet/models/roi_extractors/single_level.py:11-107 for (K, 6) rotated RoIs --
level by sqrt(w*h) (the reference's sqrt((x2-x1+1)(y2-y1+1)), :69-73, for the
box's own extent), optional roi_rescale (:75-87) of w and h.
"""
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from .rroi_align import RRoIAlign, rroi_align


class RoIAlign(nn.Module):

    def __init__(self, out_size, spatial_scale, sample_num=0, use_torchvision=False, aligned=False):
        super(RoIAlign, self).__init__()
        self.out_size = _pair(out_size)
        self.spatial_scale = float(spatial_scale)
        self.aligned = aligned
        self.sample_num = int(sample_num)
        self.use_torchvision = use_torchvision
        assert not use_torchvision, 'torchvision is not used on this path'

    def forward(self, features, rois):
        """features: NCHW; rois: (K, 5) [batch_index, x1, y1, x2, y2]."""
        assert rois.dim() == 2 and rois.size(1) == 5
        b, x1, y1, x2, y2 = rois.unbind(1)
        zero = torch.zeros_like(x1)
        if not self.aligned:   # legacy V1
            rr = torch.stack([b, (x1 + x2 + 1) * 0.5, (y1 + y2 + 1) * 0.5, x2 - x1 + 1, y2 - y1 + 1,
                              zero], 1)
        else:                  # V2, aligned=True
            off = 0.5 / self.spatial_scale
            rr = torch.stack([b, (x1 + x2) * 0.5 - off, (y1 + y2) * 0.5 - off, x2 - x1, y2 - y1,
                              zero], 1)
        return rroi_align(features, rr, self.out_size, self.spatial_scale, self.sample_num, 0.0)

    def __repr__(self):
        return '{}(out_size={}, spatial_scale={}, sample_num={}, use_torchvision={}, aligned={})'.format(
            self.__class__.__name__, self.out_size, self.spatial_scale, self.sample_num,
            self.use_torchvision, self.aligned)


class RotatedSingleRoIExtractor(nn.Module):
    """Extract rotated RoI features from a single level chosen by RoI scale."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super(RotatedSingleRoIExtractor, self).__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type', 'RRoIAlign')
        assert layer_type == 'RRoIAlign', 'rotated extractor needs the RRoIAlign layer'
        # built like single_level.py:45-52: layer_cls(spatial_scale=1/s, **cfg)
        self.roi_layers = nn.ModuleList([RRoIAlign(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def map_roi_levels(self, rois, num_levels):
        """rois (k, 6) [b, cx, cy, w, h, theta] -> level index (k,), single_level.py:54-73
        with scale = sqrt(w*h)."""
        scale = torch.sqrt(rois[:, 3] * rois[:, 4])
        target_lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def roi_rescale(self, rois, scale_factor):
        new = rois.clone()
        new[:, 3] = rois[:, 3] * scale_factor
        new[:, 4] = rois[:, 4] * scale_factor
        return new

    def forward(self, feats, rois, roi_scale_factor=None):
        feats = [f.float() for f in feats]   # force_fp32(apply_to=('feats',)), single_level.py:89
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        out_size = self.roi_layers[0].out_size
        num_levels = len(feats)
        target_lvls = self.map_roi_levels(rois, num_levels)
        roi_feats = feats[0].new_zeros(rois.size(0), self.out_channels, *out_size)
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        for i in range(num_levels):
            inds = target_lvls == i
            if inds.any():
                roi_feats[inds] = self.roi_layers[i](feats[i], rois[inds, :])
        return roi_feats
