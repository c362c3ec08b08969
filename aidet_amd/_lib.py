"""ctypes binding of the C ABI in include/obbops.h (libobbops.so).

This is the stub INTEGRATION.md describes: the reference imports its pybind
modules next to the wrappers (mmdet/ops/nms/nms_wrapper.py:4); here the flat C
ABI is loaded once and called with raw device pointers + the current HIP stream.
PyTorch is only the owner of device memory and streams.

No fallback of any kind: a missing library or GPU raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OBBOPS_LIB: an alternative build of the library (A/B measurements)
LIB_PATH = os.environ.get('OBBOPS_LIB') or os.path.join(_HERE, 'lib', 'libobbops.so')

OBB_OK = 0
OBB_EINVAL = -1
OBB_EWORKSPACE = -2
OBB_ERETRY = -4
OBB_ESTALLED = -5
OBB_EHIP = -3

KIND_RBOX = 0
KIND_POLY = 1
KIND_HBB = 2   # NMS only: x1, y1, x2, y2 (+ score) with the reference's +1 convention
MODE = {'iou': 0, 'iof': 1}
NMS_STRICT = 1          # OBB_NMS_STRICT: suppress on `>` (nms_kernel.cu:61) instead of `>=` (nms_cpu.cpp:56)
NMS_RETAIN = 2          # OBB_NMS_RETAIN: the workspace is the caller's alone from call to call -- three launches instead of four (obbops.h)
NMS_RESULT_BYTES = 64   # OBB_NMS_RESULT_BYTES
# obb_set_tuning knobs (test / experiment hooks, include/obbops.h)
(TUNE_NMS_FUSED_MAX, TUNE_NMS_DIRECT_MAX, TUNE_NMS_SCAN_BLOCKS, TUNE_RROI_GROUPS, TUNE_OVERLAPS_ROWS, TUNE_NMS_INLINE_MAX,
 TUNE_RROI_ORDER_MIN, TUNE_RROI_GATHER_CHAN, TUNE_RPN_BARRIER_TICKS, TUNE_SOFT_NMS_REG, TUNE_NMS_FUSED_COMPACT, TUNE_RROI_PLAN, TUNE_NMS_GRID,
 TUNE_SOFT_NMS_SHAPE, TUNE_RROI_NHWC_NQ, TUNE_RROI_NHWC_XCD, TUNE_RROI_NHWC_WAVES, TUNE_NMS_WATCH_RECORD, TUNE_RROI_NHWC_GATHER_WAVES,
 TUNE_NMS_RESOLVE_TAIL, TUNE_NMS_RESOLVE_PARTS, TUNE_NMS_RETAIN) = range(22)

_lib = None


class ObbOpsError(RuntimeError):
    """A libobbops call failed (mirrors the RuntimeError raised by AT_CHECK in the
    reference's C++ ops, mmdet/ops/nms/src/nms_cpu.cpp:7)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'aidet_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` or `make -C aidet_amd/csrc` (hipcc, gfx950). There is no CPU '
                'fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.obb_version.restype = i32
        L.obb_build_info.restype = ctypes.c_char_p
        L.obb_last_error.restype = ctypes.c_char_p
        L.obb_overlaps_f32.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp, vp]
        L.obb_overlaps_f32.restype = i32
        L.obb_hbb_overlaps_f64.argtypes = [vp, i64, vp, i64, i32, i32, vp, vp]
        L.obb_hbb_overlaps_f64.restype = i32
        L.obb_nms_workspace_bytes.argtypes = [i64, i32, i64]
        L.obb_nms_workspace_bytes.restype = ctypes.c_size_t
        L.obb_nms_f32.argtypes = [vp, i64, i32, f32, vp, ctypes.POINTER(i64), vp,
                                  ctypes.c_size_t, vp]
        L.obb_nms_f32.restype = i32
        L.obb_nms_grouped_f32.argtypes = [vp, i64, i32, f32, vp, vp, vp, ctypes.POINTER(i64), vp,
                                          ctypes.c_size_t, vp]
        L.obb_nms_grouped_f32.restype = i32
        L.obb_nms_ex_f32.argtypes = [vp, i64, i32, f32, vp, vp, i32, vp, ctypes.POINTER(i64), vp,
                                     ctypes.c_size_t, vp]
        L.obb_nms_ex_f32.restype = i32
        L.obb_nms_hbb_f64.argtypes = [vp, i64, f32, vp, vp, i32, vp, ctypes.POINTER(i64), vp, ctypes.c_size_t, vp]
        L.obb_nms_hbb_f64.restype = i32
        L.obb_nms_async_f32.argtypes = [vp, i64, i32, f32, vp, vp, i32, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_nms_async_f32.restype = i32
        L.obb_nms_async_result.argtypes = [vp, i64, ctypes.c_size_t, ctypes.POINTER(i64)]
        L.obb_nms_async_result.restype = i32
        L.obb_soft_nms_workspace_bytes.argtypes = [i64]
        L.obb_soft_nms_workspace_bytes.restype = ctypes.c_size_t
        L.obb_soft_nms_f32.argtypes = [vp, i64, f32, i32, f32, f32, vp, vp, ctypes.POINTER(i64), vp,
                                       ctypes.c_size_t, vp]
        L.obb_soft_nms_f32.restype = i32
        L.obb_soft_nms_f64.argtypes = [vp, i64, f32, i32, f32, f32, vp, vp, ctypes.POINTER(i64), vp]
        L.obb_soft_nms_f64.restype = i32
        L.obb_soft_nms_grouped_f32.argtypes = [vp, i64, vp, i64, f32, i32, f32, f32, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_soft_nms_grouped_f32.restype = i32
        L.obb_soft_nms_grouped_max_f32.argtypes = [vp, i64, vp, i64, i64, f32, i32, f32, f32, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_soft_nms_grouped_max_f32.restype = i32
        L.obb_rpn_select_decode_f32.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32),
                                                ctypes.POINTER(i32), i32, i32, i32, i32, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                                f32, i32, i32, vp, vp, ctypes.POINTER(i64), vp]
        L.obb_rpn_select_decode_f32.restype = i32
        L.obb_rpn_select_decode_ws_f32.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32),
                                                   ctypes.POINTER(i32), i32, i32, i32, i32, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                                   f32, i32, i32, vp, vp, ctypes.POINTER(i64), vp, ctypes.c_size_t, vp]
        L.obb_rpn_select_decode_ws_f32.restype = i32
        L.obb_rpn_select_workspace_bytes.argtypes = [ctypes.POINTER(i32), ctypes.POINTER(i32), i32, i32, i32]
        L.obb_rpn_select_workspace_bytes.restype = ctypes.c_size_t
        L.obb_rows_topk_f32.argtypes = [vp, i64, i32, i32, vp, i64, ctypes.POINTER(i32), i32, i32, i32, vp, vp, vp, vp]
        L.obb_rows_topk_f32.restype = i32
        L.obb_head_select_decode_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, f32, ctypes.POINTER(f32), ctypes.POINTER(f32), f32,
                                                 i32, i32, f32, vp, vp, vp, vp, vp]
        L.obb_head_select_decode_f32.restype = i32
        L.obb_head_gather_decode_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, i64, ctypes.POINTER(f32), ctypes.POINTER(f32),
                                                 f32, i32, i32, f32, vp, vp]
        L.obb_head_gather_decode_f32.restype = i32
        L.obb_nms_last_route.restype = i32
        L.obb_nms_last_retained.restype = i32
        L.obb_nms_retain_forget.argtypes = [vp]
        L.obb_nms_retain_forget.restype = i32
        L.obb_soft_nms_last_path.restype = i32
        L.obb_soft_nms_last_path.argtypes = []
        L.obb_set_tuning.argtypes = [i32, i64]
        L.obb_set_tuning.restype = i32
        rroi = [vp, i32, i32, i32, i32, vp, i64, i32, i32, f32, i32, f32, vp, vp]
        L.obb_rroi_align_forward_f32.argtypes = rroi
        L.obb_rroi_align_forward_f32.restype = i32
        L.obb_rroi_align_backward_f32.argtypes = rroi
        L.obb_rroi_align_backward_f32.restype = i32
        roi = [vp, i32, i32, i32, i32, vp, i64, i32, i32, f32, i32, i32, vp, vp]
        L.obb_roi_align_forward_f32.argtypes = roi
        L.obb_roi_align_forward_f32.restype = i32
        L.obb_roi_align_backward_f32.argtypes = roi
        L.obb_roi_align_backward_f32.restype = i32
        L.obb_roi_align_levels_forward_f32.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                                       ctypes.POINTER(f32), i32, i32, i32, vp, vp, i64, i32, i32,
                                                       i32, i32, f32, vp, vp]
        L.obb_roi_align_levels_forward_f32.restype = i32
        L.obb_roi_map_levels_i32.argtypes = [vp, i64, i32, f32, i32, vp, vp]
        L.obb_roi_map_levels_i32.restype = i32
        L.obb_roi_forward_workspace_bytes.argtypes = [i64]
        L.obb_roi_forward_workspace_bytes.restype = ctypes.c_size_t
        L.obb_roi_align_forward_ws_f32.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                                   ctypes.POINTER(f32), i32, i32, i32, vp, vp, i64, i32, i32,
                                                   i32, i32, f32, vp, vp, ctypes.c_size_t, vp]
        L.obb_roi_align_forward_ws_f32.restype = i32
        L.obb_roi_align_forward_nhwc_f32.argtypes = L.obb_roi_align_forward_ws_f32.argtypes
        L.obb_roi_align_forward_nhwc_f32.restype = i32
        L.obb_rroi_align_forward_nhwc_f32.argtypes = rroi
        L.obb_rroi_align_forward_nhwc_f32.restype = i32
        L.obb_nms_profile_enable.argtypes = [i32]
        L.obb_nms_profile_enable.restype = i32
        L.obb_nms_profile_last.argtypes = [ctypes.POINTER(f32)]
        L.obb_nms_profile_last.restype = i32
        L.obb_rbox_to_poly_f32.argtypes = [vp, i64, vp, vp]
        L.obb_rbox_to_poly_f32.restype = i32
        L.obb_roi_backward_workspace_bytes.argtypes = [i64]
        L.obb_roi_backward_workspace_bytes.restype = ctypes.c_size_t
        L.obb_roi_backward_gather_f32.argtypes = [vp, i32, i32, i32, i32, vp, i64, i32, i32, i32, f32, i32, f32, vp, vp,
                                                  ctypes.c_size_t, vp]
        L.obb_roi_backward_gather_f32.restype = i32
        L.obb_roi_backward_nhwc_workspace_bytes.argtypes = [i64, i32, i32, i32]
        L.obb_roi_backward_nhwc_workspace_bytes.restype = ctypes.c_size_t
        L.obb_roi_backward_gather_nhwc_f32.argtypes = L.obb_roi_backward_gather_f32.argtypes
        L.obb_roi_backward_gather_nhwc_f32.restype = i32
        L.obb_patch_groups_i32.argtypes = [vp, vp, i64, i64, vp, vp]
        L.obb_patch_groups_i32.restype = i32
        L.obb_patch_rows_to_image_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, i64, f32, vp, vp]
        L.obb_patch_rows_to_image_f32.restype = i32
        L.obb_merge_rows_split_f32.argtypes = [vp, i64, i32, i64, vp, f32, vp, vp, vp, vp]
        L.obb_merge_rows_split_f32.restype = i32
        L.obb_merge_rows_split_ld_f32.argtypes = [vp, i64, i32, i32, i64, vp, f32, vp, vp, vp, vp]
        L.obb_merge_rows_split_ld_f32.restype = i32
        L.obb_patch_rows_to_image_keyed_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, i64, f32, vp, vp, vp]
        L.obb_patch_rows_to_image_keyed_f32.restype = i32
        L.obb_gather_rows_f32.argtypes = [vp, i64, i32, vp, i64, i32, vp, vp]
        L.obb_gather_rows_f32.restype = i32
        L.obb_rows_route_workspace_bytes.argtypes = [i64, i32]
        L.obb_rows_route_workspace_bytes.restype = ctypes.c_size_t
        L.obb_rows_route_f32.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_rows_route_f32.restype = i32
        L.obb_rows_restore_workspace_bytes.argtypes = [i32, i64]
        L.obb_rows_restore_workspace_bytes.restype = ctypes.c_size_t
        L.obb_rows_restore_f32.argtypes = [vp, i64, i32, i32, ctypes.POINTER(ctypes.c_int64), i32, i64, i32, i64, vp, vp, vp,
                                           ctypes.c_size_t, vp]
        L.obb_rows_restore_f32.restype = i32
        L.obb_tiled_local_workspace_bytes.argtypes = [i64, i32, i64]
        L.obb_tiled_local_workspace_bytes.restype = ctypes.c_size_t
        L.obb_tiled_local_f32.argtypes = [vp, vp, vp, i64, vp, vp, i64, i64, f32, f32, i32, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_tiled_local_f32.restype = i32
        L.obb_tiled_merge_workspace_bytes.argtypes = [i64, i32, i32, i64, i64]
        L.obb_tiled_merge_workspace_bytes.restype = ctypes.c_size_t
        L.obb_tiled_merge_f32.argtypes = [vp, i64, i32, i32, ctypes.POINTER(ctypes.c_int64), i32, i64, i64, i64, vp, f32, i32, vp,
                                          vp, i32, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.obb_tiled_merge_f32.restype = i32
        L.obb_tiled_result.argtypes = [vp, i32, i64, i32, i32, i64, ctypes.c_size_t, ctypes.POINTER(i64)]
        L.obb_tiled_result.restype = i32
        _lib = L
    return _lib


def set_tuning(knob, value):
    """Test / experiment hook: force a launch-geometry or routing knob (value < 0 = default)."""
    check(lib().obb_set_tuning(int(knob), int(value)))


def check(rc):
    if rc != OBB_OK:
        raise ObbOpsError('libobbops error %d: %s' % (rc, lib().obb_last_error().decode()))


_gpu_ok = False


def require_gpu():
    global _gpu_ok
    if _gpu_ok:                      # (checked once: these ops are called thousands of times)
        return
    if not torch.cuda.is_available():
        raise RuntimeError('aidet_amd ops run on a HIP device only (no CPU path); '
                           'torch.cuda.is_available() is False')
    _gpu_ok = True


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


class on_device(object):
    """``with on_device(dev):`` -- torch's current device = dev (kernels launch there).  Unlike ``torch.cuda.device`` it
    does nothing when the device already is current (the usual case: ~4 us per call saved, and every operator of this
    package makes several calls)."""

    def __init__(self, dev):
        self.dev, self.prev = dev, None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev.index and self.dev.index is not None:
            self.prev = cur
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


def current_stream_handle(device):
    """hipStream_t of torch's current stream on `device` as an int."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device):
    return ctypes.c_void_p(current_stream_handle(device))


def to_device_f32(t, device_id=None):
    """Contiguous fp32 copy of `t` on a HIP device (the input's own device for a
    HIP tensor, else cuda:{device_id or current})."""
    require_gpu()
    if t.is_cuda:
        dev = t.device
    else:
        dev = torch.device('cuda', torch.cuda.current_device() if device_id is None else device_id)
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()
