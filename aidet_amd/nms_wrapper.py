"""Rotated / polygon NMS with the calling conventions of the reference's
``mmdet/ops/nms/nms_wrapper.py`` (``nms`` at :7-60), so that
``getattr(nms_wrapper, nms_cfg['type'])(cls_dets, **cfg)`` -- the selection the
reference does in mmdet/core/post_processing/bbox_nms.py:60-64 and
rbbox_nms.py:25-27 -- works with ``type='rnms'`` / ``'poly_nms'`` /
``'thetaobb_nms'`` (the name the reference's disabled call expects,
rbbox_nms.py:97).

All compute is the HIP path ``obb_nms_f32`` (include/obbops.h); there is no CPU
implementation.  CPU tensors / numpy arrays are moved to the GPU and the result
is returned in the caller's container type and device, as the reference does
for ``device_id`` (nms_wrapper.py:41-43).
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib


# Workspace cache: the call is synchronous on its stream (it returns the keep count),
# so one buffer per (thread, device, stream) can be reused by the next call there.
# (ctypes releases the GIL during the call, hence per thread.)
_tls = threading.local()
_ws_bytes = {}


def _workspace(L, dev, stream, n, kind, capacity):
    key = (n, kind, capacity)
    nbytes = _ws_bytes.get(key)
    if nbytes is None:
        nbytes = L.obb_nms_workspace_bytes(n, kind, capacity)
        if nbytes == 0:
            raise _lib.ObbOpsError('obb_nms_workspace_bytes failed: %s' %
                                   L.obb_last_error().decode())
        if len(_ws_bytes) > 4096:
            _ws_bytes.clear()
        _ws_bytes[key] = nbytes
    cache = getattr(_tls, 'ws', None)
    if cache is None:
        cache = _tls.ws = {}
    ckey = (dev.index, stream)
    ws = cache.get(ckey)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        cache[ckey] = ws
    return ws, nbytes


def _run_nms(dets_dev, kind, iou_thr, groups=None, thrs=None):
    """dets_dev: contiguous fp32 HIP tensor (n, 6|9). Returns int64 keep (K,).

    groups: optional int tensor (n,) -- only boxes of equal group interact.
    thrs: optional float tensor (n,) -- per-box threshold (the earlier box's applies).
    """
    n = dets_dev.size(0)
    dev = dets_dev.device
    L = _lib.lib()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    info = getattr(_tls, 'info', None)      # (argtypes convert plain ints: no c_void_p objects per call)
    if info is None:
        info = _tls.info = (ctypes.c_int64 * 4)()
    gptr = tptr = None
    if groups is not None:
        groups = groups.detach().to(device=dev, dtype=torch.int32).contiguous()
        assert groups.numel() == n
        gptr = groups.data_ptr()
    if thrs is not None:
        thrs = thrs.detach().to(device=dev, dtype=torch.float32).contiguous()
        assert thrs.numel() == n
        tptr = thrs.data_ptr()
    cur = torch.cuda.current_device()
    same_dev = cur == dev.index
    if not same_dev:
        torch.cuda.set_device(dev)
    try:
        stream = _lib.current_stream_handle(dev)
        capacity = 0  # library default
        for _ in range(3):
            ws, nbytes = _workspace(L, dev, stream, n, kind, capacity)
            rc = L.obb_nms_grouped_f32(dets_dev.data_ptr(), n, kind, iou_thr, gptr, tptr,
                                       keep.data_ptr(), info, ws.data_ptr(), nbytes, stream)
            if rc != _lib.OBB_EWORKSPACE:
                break
            capacity = int(info[3])  # a pair capacity the library says suffices
    finally:
        if not same_dev:
            torch.cuda.set_device(cur)
    if rc != _lib.OBB_OK:
        _lib.check(rc)
    return keep[:info[0]], (info[1], info[2], info[3])


def _as_tensor(x):
    if x is None or isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x))


def _nms_generic(dets, iou_thr, device_id, cols, kind, name, groups=None, thrs=None):
    # container handling mirrors nms_wrapper.py:36-47
    if isinstance(dets, torch.Tensor):
        is_numpy = False
        dets_th = dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    if dets_th.dim() != 2 or dets_th.size(1) != cols:
        raise ValueError('{} expects dets of shape (n, {}), got {}'.format(
            name, cols, tuple(dets_th.shape)))

    if dets_th.shape[0] == 0:  # nms_wrapper.py:50-51
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        d = _lib.to_device_f32(dets_th, device_id)
        inds, _ = _run_nms(d, kind, float(iou_thr), _as_tensor(groups), _as_tensor(thrs))
        if not is_numpy:
            inds = inds.to(dets_th.device)  # keep lives where the input lives
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds


def rnms(dets, iou_thr, device_id=None, groups=None, thrs=None):
    """Greedy NMS on theta-OBBs.

    Arguments:
        dets (torch.Tensor or np.ndarray): (n, 6) <cx, cy, w, h, theta, score>.
        iou_thr (float): a box is suppressed by a higher-scored kept box when
            their rotated IoU is >= iou_thr (the CPU reference's comparison,
            mmdet/ops/nms/src/nms_cpu.cpp:56).
        device_id (int, optional): HIP device for numpy / CPU input.
        groups (int array, optional): (n,) group id per box; boxes of different
            groups never suppress each other (per-class / per-image NMS in one
            call -- what bbox_nms.py:56-63 does with coordinate offsets).
        thrs (float array, optional): (n,) per-box threshold replacing iou_thr
            (per-class thresholds of the DOTA merge, dota.py:321-331).

    Returns:
        tuple: kept dets and int64 indices (ascending, nms_cpu.cpp:59), same
        container type as the input.

    Example:
        >>> dets = np.array([[50., 34., 2., 3.5, 0., 0.9],
        >>>                  [50.1, 34.1, 1.8, 2.5, 0., 0.9],
        >>>                  [37.6, 13., 4.6, 3., 0., 0.4],
        >>>                  [37.4, 13.7, 4.5, 4., 0., 0.3]], dtype=np.float32)
        >>> kept, inds = rnms(dets, 0.7)
        >>> assert len(inds) == len(kept) == 4
    """
    return _nms_generic(dets, iou_thr, device_id, 6, _lib.KIND_RBOX, 'rnms', groups, thrs)


def poly_nms(dets, iou_thr, device_id=None, groups=None, thrs=None):
    """Greedy NMS on convex quadrilaterals: dets (n, 9) <x1,y1,..,x4,y4,score>
    (the 8-point polygons the reference writes for the DOTA merge,
    mmdet/datasets/dota.py:197-235, merged by the external mergebypoly)."""
    return _nms_generic(dets, iou_thr, device_id, 9, _lib.KIND_POLY, 'poly_nms', groups, thrs)


# the names the reference's own (disabled) call sites use
thetaobb_nms = rnms      # mmdet/core/post_processing/rbbox_nms.py:97
pointobb_nms = poly_nms


def nms(dets, iou_thr, device_id=None, groups=None):
    """Horizontal-box NMS with the reference's exact signature
    (nms_wrapper.py:7), served by the rotated kernels through the theta = 0
    mapping  w = x2-x1+1, h = y2-y1+1, centre = box centre  (the continuous box
    [x1-.5, x2+.5] x [y1-.5, y2+.5] has the reference's +1 intersection widths,
    nms_cpu.cpp:19,52-55).  dets: (n, 5) <x1, y1, x2, y2, score>."""
    if isinstance(dets, torch.Tensor):
        is_numpy = False
        dets_th = dets
    elif isinstance(dets, np.ndarray):
        is_numpy = True
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    if dets_th.dim() != 2 or dets_th.size(1) != 5:
        raise ValueError('nms expects dets of shape (n, 5), got {}'.format(tuple(dets_th.shape)))
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        d = _lib.to_device_f32(dets_th, device_id)
        # the kernel reads x1, y1, x2, y2 and builds the theta = 0 OBB itself (OBB_KIND_HBB)
        inds, _ = _run_nms(d, _lib.KIND_HBB, float(iou_thr), _as_tensor(groups))
        if not is_numpy:
            inds = inds.to(dets_th.device)
    if is_numpy:
        inds = inds.cpu().numpy()
    return dets[inds, :], inds
