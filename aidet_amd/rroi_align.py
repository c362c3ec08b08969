"""Rotated RoI-Align: autograd Function + nn.Module with the construction and
call conventions of the reference's RoIAlign (mmdet/ops/roi_align/roi_align.py:9-152)
so that ``SingleRoIExtractor`` can build it from a config
(``roi_layer=dict(type='RRoIAlign', out_size=7, sample_num=2)``,
mmdet/models/roi_extractors/single_level.py:45-52) and read ``.out_size``
(single_level.py:94).  Compute: ``obb_rroi_align_{forward,backward}_f32``.
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib


class RRoIAlignFunction(Function):

    @staticmethod
    def forward(ctx, features, rois, out_size, spatial_scale, sample_num=0, min_size=1.0):
        out_h, out_w = _pair(out_size)
        assert isinstance(out_h, int) and isinstance(out_w, int)
        if not features.is_cuda:
            # same contract as the reference op (roi_align.py:41-42)
            raise NotImplementedError('RRoIAlign runs on a HIP device only')
        assert rois.dim() == 2 and rois.size(1) == 6, 'rois must be (K, 6) [b, cx, cy, w, h, theta]'
        ctx.spatial_scale = float(spatial_scale)
        ctx.sample_num = int(sample_num)
        ctx.min_size = float(min_size)
        ctx.out_hw = (out_h, out_w)
        ctx.feature_size = features.size()
        ctx.in_dtype = features.dtype

        from .roi_layers import map_layout, roi_forward
        layout, (feat,) = map_layout([features])
        ctx.layout = layout
        r = rois.detach().to(device=feat.device, dtype=torch.float32).contiguous()
        ctx.save_for_backward(r)
        out = roi_forward([feat], r, None, out_h, out_w, [ctx.spatial_scale], ctx.sample_num, 0, ctx.min_size, layout)
        return out.to(features.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (r,) = ctx.saved_tensors
        B, C, H, W = ctx.feature_size
        out_h, out_w = ctx.out_hw
        assert grad_output.is_cuda
        grad_input = None
        if ctx.needs_input_grad[0]:
            from .roi_layers import roi_backward
            grad_input = roi_backward(grad_output, r, (B, C, H, W), 0, out_h, out_w, ctx.spatial_scale,
                                      ctx.sample_num, ctx.min_size, ctx.layout).to(ctx.in_dtype)
        # gradient w.r.t. features only, like roi_align.py:46-73
        return grad_input, None, None, None, None, None


rroi_align = RRoIAlignFunction.apply


class RRoIAlign(nn.Module):

    def __init__(self, out_size, spatial_scale, sample_num=0, min_size=1.0):
        """
        Args:
            out_size (int or tuple): h, w of the pooled output
            spatial_scale (float): scale the RoI centre and size by this number
            sample_num (int): samples per bin and axis; 0 = ceil(bin size)
            min_size (float): lower clamp of the scaled RoI width/height
                (1 = RoIAlign V2 aligned=False, 0 = legacy V1;
                roi_align_kernel_v2.cu:87-90, roi_align_kernel.cu:85-86)
        """
        super(RRoIAlign, self).__init__()
        self.out_size = _pair(out_size)
        self.spatial_scale = float(spatial_scale)
        self.sample_num = int(sample_num)
        self.min_size = float(min_size)

    def forward(self, features, rois):
        """
        Args:
            features: NCHW feature map
            rois: Kx6 boxes: batch index, then cx, cy, w, h, theta (image scale)
        """
        assert rois.dim() == 2 and rois.size(1) == 6
        return rroi_align(features, rois, self.out_size, self.spatial_scale, self.sample_num,
                          self.min_size)

    def __repr__(self):
        format_str = self.__class__.__name__
        format_str += '(out_size={}, spatial_scale={}, sample_num={}, min_size={})'.format(
            self.out_size, self.spatial_scale, self.sample_num, self.min_size)
        return format_str
