"""RoI layers built on the rotated RoI-Align kernel.

``RoIAlign``: drop-in for the reference's horizontal op
(mmdet/ops/roi_align/roi_align.py:79-152; its CUDA kernels need THC headers and
cannot be built on ROCm): V1 (``aligned=False`` in the reference means the LEGACY
kernel, roi_align.py:28-36 -> roi_align_kernel.cu:64-117) and V2
(``aligned=True``, roi_align_kernel_v2.cu:62-128) are both served by the rotated
kernels at theta = 0 (``obb_roi_align_*`` decodes the (K, 5) RoI row in the kernel):

  V1:  start = x1*s, end = (x2+1)*s, size clamped at 0
       == rotated RoI  centre ((x1+x2+1)/2, (y1+y2+1)/2), size (x2-x1+1, y2-y1+1), min_size 0
  V2 aligned:  start = x1*s - 0.5, size (x2-x1)*s unclamped
       == rotated RoI  centre ((x1+x2)/2 - 0.5/s, ...), size (x2-x1, y2-y1), min_size 0

``RotatedSingleRoIExtractor``: the multi-level extractor of
mmdet/models/roi_extractors/single_level.py:11-107 for (K, 6) rotated RoIs --
level by sqrt(w*h) (the reference's sqrt((x2-x1+1)(y2-y1+1)), :69-73, for the
box's own extent), optional roi_rescale (:75-87) of w and h.
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair


_warned_layout = set()


def nchw_f32(features):
    """The (B, C, H, W) fp32 contiguous view the NCHW kernels read.  A map that is not already that -- a half-precision
    map, a strided slice, a channels-last map the channels-last kernels cannot take -- is COPIED (67 MB at configs[2]);
    the reference's kernels read `features.contiguous()` the same way (roi_align_cuda.cpp:44, CHECK_CONTIGUOUS), but
    silent costs are easy to miss: say so once per cause."""
    f = features.detach()
    why = None
    if f.dtype != torch.float32:
        why = 'dtype %s' % str(f.dtype).replace('torch.', '')
    elif not f.is_contiguous():
        why = 'channels-last layout' if f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) else 'non-contiguous strides'
    if why is not None and why not in _warned_layout:
        _warned_layout.add(why)
        import warnings
        warnings.warn('RoI-Align: the feature map (%s) is copied to contiguous NCHW fp32 before pooling (%.1f MB per call)'
                      % (why, f.numel() * 4 / 1e6), UserWarning, stacklevel=3)
    return f.to(torch.float32).contiguous()


# Channels-last maps are pooled IN PLACE by the channels-last kernels (obb_roi_align_forward_nhwc_f32, round 5): a tap
# of all channels is one contiguous run there.  False: every map is brought to NCHW first (rounds 1-4).
NHWC_NATIVE = True


def is_nhwc(f):
    """a (B, C, H, W) fp32 HIP tensor whose MEMORY is (B, H, W, C) and which obb_roi_align_forward_nhwc_f32 accepts"""
    return (NHWC_NATIVE and f.dim() == 4 and f.dtype == torch.float32 and f.is_cuda and f.size(1) % 4 == 0
            and f.size(1) > 1 and f.size(2) * f.size(3) > 1           # (C == 1 or one pixel: both layouts are the same memory)
            and not f.is_contiguous() and f.is_contiguous(memory_format=torch.channels_last)
            and f.size(1) * f.size(2) * f.size(3) < (1 << 30) and f.data_ptr() % 16 == 0)


def map_layout(feats):
    """('nhwc', maps as they are) when EVERY level is channels-last memory the kernels take, else ('nchw', contiguous
    NCHW fp32 maps; whatever is not that already is copied, with a warning)."""
    feats = [f.detach() for f in feats]
    if all(is_nhwc(f) for f in feats):
        return 'nhwc', feats
    return 'nchw', [nchw_f32(f) for f in feats]


from . import _lib
from .rroi_align import RRoIAlign, rroi_align  # noqa: F401  (rroi_align: the mapping the kernel now does itself)


# Which backward runs: 'auto' | 'gather' (obb_roi_backward_gather_f32: plan per RoI + one block per map tile; no
# atomics, no zero fill, bit-reproducible) | 'atomics' (the scatter kernels behind obb_(r)roi_align_backward_f32).
# Measured on MI355X (tools/exp_bwd_crossover.py, rotated RoIs on a 256^2 x 256-channel map; scatter incl. its zero fill):
# 16 RoIs 0.034 / 0.039 ms (gather / scatter), 64: 0.037 / 0.052, 512: 0.071 / 0.193, 1024: 0.112 / 0.361; config 4's
# first level (2000 horizontal RoIs covering the map 22x): 0.35 / 1.67 ms -- the scatter pays per atomic request and for
# its direct-atomics tail, the gather per (tile, RoI) pair and entry (round 3: merged taps, pipelined requests).  'auto' =
# gather from GATHER_MIN_ROIS RoIs on (below, the scatter's smaller fixed cost on small maps).
BACKWARD = 'auto'
GATHER_MIN_ROIS = 16


def roi_backward(grad_out, rois, feature_size, roi_kind, out_h, out_w, spatial_scale, sample_num, min_size, layout='nchw'):
    """Gradient of any RoI-Align flavour w.r.t. the feature map: (K, C, ph, pw) -> (B, C, H, W), fp32.

    ``roi_kind`` 0: rotated (K, 6) rows, 1: RoIAlign V1, 2: V2-aligned (K, 5) rows.  ``roi_layers.BACKWARD``
    selects the formulation (see there).  ``layout`` 'nhwc' (the forward pooled a channels-last map): the gradient comes
    back as a channels-last tensor too, written in place by ``obb_roi_backward_gather_nhwc_f32`` (C % 4 == 0, <= 65535 RoIs;
    anything else takes the NCHW route and the caller's layout conversion)."""
    B, C, H, W = feature_size
    g = grad_out.detach().to(torch.float32).contiguous()
    K = rois.size(0)
    dev = g.device
    L = _lib.lib()
    with _lib.on_device(dev):
        st = _lib.current_stream_handle(dev)
        if layout == 'nhwc' and C % 4 == 0 and K <= 65535 and BACKWARD != 'atomics':
            grad = torch.empty((B, C, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            ws = torch.empty(max(int(L.obb_roi_backward_nhwc_workspace_bytes(K, C, out_h, out_w)), 16), dtype=torch.uint8, device=dev)
            _lib.check(L.obb_roi_backward_gather_nhwc_f32(g.data_ptr(), B, C, H, W, rois.data_ptr(), K, roi_kind, out_h, out_w,
                                                          spatial_scale, sample_num, min_size, grad.data_ptr(), ws.data_ptr(),
                                                          ws.numel(), st))
            return grad
        if BACKWARD == 'gather' or (BACKWARD == 'auto' and K >= GATHER_MIN_ROIS):
            grad = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
            ws = torch.empty(max(int(L.obb_roi_backward_workspace_bytes(K)), 16), dtype=torch.uint8, device=dev)
            _lib.check(L.obb_roi_backward_gather_f32(g.data_ptr(), B, C, H, W, rois.data_ptr(), K, roi_kind, out_h, out_w,
                                                     spatial_scale, sample_num, min_size, grad.data_ptr(), ws.data_ptr(),
                                                     ws.numel(), st))
            return grad
        grad = torch.zeros((B, C, H, W), dtype=torch.float32, device=dev)
        if K > 0 and C > 0:
            if roi_kind == 0:
                _lib.check(L.obb_rroi_align_backward_f32(g.data_ptr(), B, C, H, W, rois.data_ptr(), K, out_h, out_w,
                                                         spatial_scale, sample_num, min_size, grad.data_ptr(), st))
            else:
                _lib.check(L.obb_roi_align_backward_f32(g.data_ptr(), B, C, H, W, rois.data_ptr(), K, out_h, out_w,
                                                        spatial_scale, sample_num, 1 if roi_kind == 2 else 0,
                                                        grad.data_ptr(), st))
    return grad


ORDER_MIN_ROIS = 1024    # = kOrderMinK of obb_rroi.hip: fewer RoIs are not worth the ordering launch


def roi_forward(feats, rois, levels, out_h, out_w, scales, sample_num, roi_kind, min_size, layout='nchw'):
    """The forward of every RoI-Align flavour: ``obb_roi_align_forward_ws_f32`` (``layout`` 'nchw') or
    ``obb_roi_align_forward_nhwc_f32`` ('nhwc': the maps' memory is (B, H, W, C)) -- one launch over the level(s) after
    one small launch that orders the RoIs along a Morton curve (L2 locality; workspace from torch's caching allocator).

    feats: list of fp32 (B, C, H_l, W_l) HIP tensors, contiguous in ``layout``; rois: contiguous fp32 (K, 5 | 6) on the
    same device; levels: int32 (K,) or None for a single map.  Returns (K, C, out_h, out_w) fp32, contiguous."""
    dev = feats[0].device
    B, C = feats[0].shape[:2]
    K, n = rois.size(0), len(feats)
    out = torch.empty((K, C, out_h, out_w), dtype=torch.float32, device=dev)
    if K == 0 or C == 0:
        return out
    L = _lib.lib()
    ptrs = (ctypes.c_void_p * n)(*[f.data_ptr() for f in feats])
    hs = (ctypes.c_int * n)(*[f.shape[2] for f in feats])
    wd = (ctypes.c_int * n)(*[f.shape[3] for f in feats])
    sc = (ctypes.c_float * n)(*[float(x) for x in scales])
    # (the library orders the RoIs from ORDER_MIN_ROIS on; below, no workspace is needed -- and this function is per-call
    # overhead of a 40 us kernel: no allocation, no context manager unless the device has to be switched)
    ws_ptr, ws_bytes = None, 0
    if K >= ORDER_MIN_ROIS:
        ws = torch.empty(int(L.obb_roi_forward_workspace_bytes(K)), dtype=torch.uint8, device=dev)      # (order + per-RoI plan records)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev)
    try:
        fn = L.obb_roi_align_forward_nhwc_f32 if layout == 'nhwc' else L.obb_roi_align_forward_ws_f32
        rc = fn(ptrs, hs, wd, sc, n, B, C, rois.data_ptr(),
                None if levels is None else levels.data_ptr(), K, out_h, out_w,
                int(sample_num), int(roi_kind), float(min_size), out.data_ptr(),
                ws_ptr, ws_bytes, _lib.current_stream_handle(dev))
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    _lib.check(rc)
    return out


class RoIAlignFunction(Function):
    """Horizontal RoIAlign (roi_align.py:9-73): features NCHW, rois (K, 5)
    [batch, x1, y1, x2, y2]; gradient w.r.t. the features only."""

    @staticmethod
    def forward(ctx, features, rois, out_size, spatial_scale, sample_num=0, aligned=False):
        out_h, out_w = _pair(out_size)
        if not features.is_cuda:
            raise NotImplementedError('RoIAlign runs on a HIP device only')   # roi_align.py:41-42
        assert rois.dim() == 2 and rois.size(1) == 5
        ctx.args = (out_h, out_w, float(spatial_scale), int(sample_num), 1 if aligned else 0)
        ctx.feature_size = features.size()
        ctx.in_dtype = features.dtype
        layout, (feat,) = map_layout([features])
        ctx.layout = layout
        r = rois.detach().to(device=feat.device, dtype=torch.float32).contiguous()
        ctx.save_for_backward(r)
        out = roi_forward([feat], r, None, out_h, out_w, [ctx.args[2]], ctx.args[3], 2 if aligned else 1, 0.0, layout)
        return out.to(features.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (r,) = ctx.saved_tensors
        B, C, H, W = ctx.feature_size
        out_h, out_w, scale, sample_num, aligned = ctx.args
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = roi_backward(grad_output, r, (B, C, H, W), 2 if aligned else 1, out_h, out_w, scale,
                                      sample_num, 0.0, ctx.layout).to(ctx.in_dtype)
        return grad_input, None, None, None, None, None


roi_align = RoIAlignFunction.apply


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
        return roi_align(features, rois, self.out_size, self.spatial_scale, self.sample_num, self.aligned)

    def __repr__(self):
        return '{}(out_size={}, spatial_scale={}, sample_num={}, use_torchvision={}, aligned={})'.format(
            self.__class__.__name__, self.out_size, self.spatial_scale, self.sample_num,
            self.use_torchvision, self.aligned)


def map_levels(rois, num_levels, finest_scale, roi_kind):
    """``map_roi_levels`` of the extractors below as ONE launch (int32 levels) for fp32 HIP RoIs; the tensor
    expressions of the methods are the specification (and every other case)."""
    r = rois.detach().contiguous()
    out = torch.empty(r.size(0), dtype=torch.int32, device=r.device)
    with _lib.on_device(r.device):
        _lib.check(_lib.lib().obb_roi_map_levels_i32(r.data_ptr(), r.size(0), int(roi_kind), float(finest_scale), int(num_levels),
                                                    out.data_ptr(), _lib.current_stream_handle(r.device)))
    return out


def pyramid_roi_align(feats, rois, levels, out_size, scales, sample_num, roi_kind, min_size=1.0):
    """RoI-Align of every RoI on the pyramid level ``levels`` names, in ONE kernel launch
    (obb_roi_align_forward_ws_f32): no per-level mask / gather / launch / scatter
    (single_level.py:101-106).  Inference only (no autograd graph).

    feats: list of (B, C, H_l, W_l) fp32 HIP tensors; rois (K, 5 | 6); levels (K,) int;
    roi_kind 0 rotated (K, 6), 1 horizontal V1, 2 horizontal V2 aligned."""
    layout, feats = map_layout(feats)
    dev = feats[0].device
    r = rois.detach().to(device=dev, dtype=torch.float32).contiguous()
    lv = levels.detach().to(device=dev, dtype=torch.int32).contiguous()
    out_h, out_w = _pair(out_size)
    return roi_forward(feats, r, lv, out_h, out_w, scales, sample_num, roi_kind, min_size, layout)


def _fusable(layers, feats, rois):
    """one launch over the pyramid: inference, HIP tensors, identical layers"""
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return False
    a = layers[0]
    return all(f.is_cuda for f in feats) and rois.is_cuda and len(feats) <= 8 and all(
        type(l) is type(a) and l.out_size == a.out_size and l.sample_num == a.sample_num and
        getattr(l, 'aligned', None) == getattr(a, 'aligned', None) and
        getattr(l, 'min_size', None) == getattr(a, 'min_size', None) for l in layers)


class SingleRoIExtractor(nn.Module):
    """``SingleRoIExtractor`` of mmdet/models/roi_extractors/single_level.py:11-107 for horizontal
    (K, 5) RoIs: every RoI is pooled from the ONE pyramid level its scale selects
    (``floor(log2(sqrt(w * h) / finest_scale + 1e-6))``, +1 pixel widths, :69-73).

    The reference builds its layers with ``getattr(ops, type)(spatial_scale=1/s, **cfg)``
    (:45-52); here ``type`` is looked up among this module's layers (RoIAlign, RRoIAlign).
    """

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super(SingleRoIExtractor, self).__init__()
        cfg = dict(roi_layer)
        layer_cls = {'RoIAlign': RoIAlign, 'RRoIAlign': RRoIAlign}[cfg.pop('type', 'RoIAlign')]
        self.roi_layers = nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def map_roi_levels(self, rois, num_levels):
        scale = torch.sqrt((rois[:, 3] - rois[:, 1] + 1) * (rois[:, 4] - rois[:, 2] + 1))
        target_lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return target_lvls.clamp(min=0, max=num_levels - 1).long()

    def roi_rescale(self, rois, scale_factor):
        """grow / shrink every RoI about its centre (single_level.py:75-87)."""
        cx, cy = (rois[:, 1] + rois[:, 3]) * 0.5, (rois[:, 2] + rois[:, 4]) * 0.5
        new_w = (rois[:, 3] - rois[:, 1] + 1) * scale_factor
        new_h = (rois[:, 4] - rois[:, 2] + 1) * scale_factor
        return torch.stack((rois[:, 0], cx - new_w * 0.5 + 0.5, cy - new_h * 0.5 + 0.5,
                            cx + new_w * 0.5 - 0.5, cy + new_h * 0.5 - 0.5), dim=-1)

    def forward(self, feats, rois, roi_scale_factor=None):
        feats = [f.float() for f in feats]   # force_fp32(apply_to=('feats',)), single_level.py:89
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        out_size = self.roi_layers[0].out_size
        fused = _fusable(self.roi_layers, feats, rois)
        target_lvls = (map_levels(rois, len(feats), self.finest_scale, 1) if fused and rois.dtype == torch.float32 and rois.size(1) == 5
                       else self.map_roi_levels(rois, len(feats)))
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        if fused:
            a = self.roi_layers[0]
            kind = 0 if isinstance(a, RRoIAlign) else (2 if a.aligned else 1)
            return pyramid_roi_align(feats, rois, target_lvls, out_size, [l.spatial_scale for l in self.roi_layers],
                                     a.sample_num, kind, getattr(a, 'min_size', 1.0))
        # (training) one pass over the levels with the RoIs grouped by level (a stable sort and ONE host
        # read of the level sizes) instead of a boolean mask, a gather and a scatter per level
        order = torch.argsort(target_lvls, stable=True)
        sizes = torch.bincount(target_lvls, minlength=len(feats)).tolist()
        sorted_rois = rois[order]
        parts, start = [], 0
        for i, n in enumerate(sizes):
            if n:
                parts.append(self.roi_layers[i](feats[i], sorted_rois[start:start + n]))
            start += n
        if not parts:
            return feats[0].new_zeros(0, self.out_channels, *out_size)
        roi_feats = torch.cat(parts, 0)
        out = torch.empty_like(roi_feats)
        out[order] = roi_feats
        return out


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
        fused = _fusable(self.roi_layers, feats, rois)
        target_lvls = (map_levels(rois, num_levels, self.finest_scale, 0) if fused and rois.dtype == torch.float32 and rois.size(1) == 6
                       else self.map_roi_levels(rois, num_levels))
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        if fused:
            a = self.roi_layers[0]
            return pyramid_roi_align(feats, rois, target_lvls, out_size, [l.spatial_scale for l in self.roi_layers],
                                     a.sample_num, 0, a.min_size)
        roi_feats = feats[0].new_zeros(rois.size(0), self.out_channels, *out_size)
        for i in range(num_levels):
            inds = target_lvls == i
            if inds.any():
                roi_feats[inds] = self.roi_layers[i](feats[i], rois[inds, :])
        return roi_feats
