"""NMS operators with the calling conventions of the reference's
``mmdet/ops/nms/nms_wrapper.py`` (``nms`` :7-60, ``soft_nms`` :63-118), so that
``getattr(nms_wrapper, nms_cfg['type'])(cls_dets, **cfg)`` -- the selection the
reference does in mmdet/core/post_processing/bbox_nms.py:60-64 and
rbbox_nms.py:25-27 -- works with ``type='nms'`` / ``'soft_nms'`` / ``'rnms'`` /
``'poly_nms'`` / ``'thetaobb_nms'`` (the name the reference's disabled call expects,
rbbox_nms.py:97).

All compute is the HIP library behind include/obbops.h; there is no CPU
implementation.  CPU tensors / numpy arrays are moved to the GPU and the result
is returned in the caller's container type and device, as the reference does
for ``device_id`` (nms_wrapper.py:41-43).

Threshold rule.  Every op suppresses on ``overlap >= iou_thr``, the rule of the CPU
reference (nms_cpu.cpp:56) that BASELINE.json names as the oracle.  The reference's own
CUDA kernel uses ``>`` (nms_kernel.cu:61); pass ``strict=True`` to get that comparison
(the two only differ for pairs whose IoU equals the threshold exactly).
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib


# Workspace cache: a call is synchronous on its stream (it returns the keep count),
# so one buffer per (thread, device, stream) can be reused by the next call there.
# (ctypes releases the GIL during the call, hence per thread.)
_tls = threading.local()
_ws_bytes = {}

# upper bound on the workspace a retry may ask for (a dense scene needs 16 B per
# overlapping pair; the library reports the pair capacity that suffices)
MAX_WORKSPACE_BYTES = 64 << 30
SOFT_NMS_F64_MAX = 4096     # obb_soft_nms_f64 keeps its state in registers


def _workspace(L, dev, stream, n, kind, capacity):
    key = (n, kind, capacity)
    # (capacity 0 = the library's default, which follows the calling thread's capacity hints: never cached here)
    nbytes = _ws_bytes.get(key) if capacity else None
    if nbytes is None:
        nbytes = L.obb_nms_workspace_bytes(n, kind, capacity)
        if nbytes == 0:
            raise _lib.ObbOpsError('obb_nms_workspace_bytes failed: %s' %
                                   L.obb_last_error().decode())
        if len(_ws_bytes) > 4096:
            _ws_bytes.clear()
        _ws_bytes[key] = nbytes
    if nbytes > MAX_WORKSPACE_BYTES:
        raise _lib.ObbOpsError(
            'NMS on %d boxes needs a %.1f GiB pair workspace (capacity %d pairs): more than '
            'nms_wrapper.MAX_WORKSPACE_BYTES; the scene is almost all mutual overlaps' %
            (n, nbytes / 2.0 ** 30, capacity))
    cache = getattr(_tls, 'ws', None)
    if cache is None:
        cache = _tls.ws = {}
    ckey = (dev.index, stream)
    ws = cache.get(ckey)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            L.obb_nms_retain_forget(ws.data_ptr())            # (OBB_NMS_RETAIN: its memory goes back to the allocator)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        cache[ckey] = ws
    return ws, nbytes


def release_workspaces():
    """Drops the calling thread's cached NMS workspaces (this module's and the ATen binding's): about 1.5 KB per box of
    the largest call plus the grid route's table (150 MB at 1e5 boxes, 2 GB at 1.25e6).  The next call allocates and
    seeds a new one."""
    for ws in getattr(_tls, 'ws', {}).values():
        _lib.lib().obb_nms_retain_forget(ws.data_ptr())       # (the memory goes back to the allocator)
    _tls.ws = {}
    ext = _fast_ext()
    if ext is not None and hasattr(ext, 'release_workspaces'):
        ext.release_workspaces()


def _side_inputs(dev, n, groups, thrs):
    gptr = tptr = None
    if groups is not None:
        groups = groups.detach().to(device=dev, dtype=torch.int32).contiguous()
        assert groups.numel() == n
        gptr = groups.data_ptr()
    if thrs is not None:
        thrs = thrs.detach().to(device=dev, dtype=torch.float32).contiguous()
        assert thrs.numel() == n
        tptr = thrs.data_ptr()
    return groups, gptr, thrs, tptr


class _OnDevice(object):
    """`with _OnDevice(dev):` -- torch's current device = dev (kernels launch there)."""

    def __init__(self, dev):
        self.dev, self.prev = dev, None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev.index:
            self.prev = cur
            torch.cuda.set_device(self.dev)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


_ext = False     # obbops_torch (the ATen-level binding): None = not built, False = not tried yet




def _fast_ext():
    """``obbops_torch`` when it is built (``__graft_entry__.build()`` does): one pybind11 call then does what the ctypes
    path below does in ~15 interpreter-level steps -- allocation of the keep list and the workspace, device guard, stream
    lookup, argument marshalling (12 us of a 110 us call at 100k boxes)."""
    global _ext
    if _ext is False:
        if os.environ.get('OBBOPS_LIB'):
            # an alternative build of the library was asked for (A/B measurements): the extension is linked against
            # the product library and would silently run THAT one
            _ext = None
            return _ext
        try:
            from . import torch_ext
            _ext = torch_ext.ext() if hasattr(torch_ext.ext(), 'nms_ex') else None
        except Exception:      # not built: the ctypes path serves everything
            _ext = None
    return _ext


def _run_nms(dets_dev, kind, iou_thr, groups=None, thrs=None, strict=False):
    """dets_dev: contiguous fp32 HIP tensor (n, 5|6|9). Returns int64 keep (K,).

    groups: optional int tensor (n,) -- only boxes of equal group interact.
    thrs: optional float tensor (n,) -- per-box threshold (the earlier box's applies).
    """
    n = dets_dev.size(0)
    dev = dets_dev.device
    ext = _fast_ext()
    if ext is not None and n > 0:
        if groups is not None or thrs is not None:
            groups, _g, thrs, _t = _side_inputs(dev, n, groups, thrs)
        flags = _lib.NMS_STRICT if strict else 0     # (the binding adds OBB_NMS_RETAIN: it owns the workspace it passes)
        # A call whose pair lists overflow is repeated with the capacity the library asks for -- twice the time.  A
        # caller that does this every time (the RPN of a dense scene: 8768 boxes, 5.7e5 overlapping pairs, 0.32 instead
        # of 0.19 ms per call) gets that capacity from the start next time: the LIBRARY keeps the hint (per calling
        # thread, kind and size class; obb_nms_workspace_bytes(n, kind, 0) follows it), so capacity 0 here means
        # "the default, or what the last overflowing call of this size class needed".
        capacity = 0
        while True:
            try:
                keep, rc, info = ext.nms_ex(dets_dev, kind, iou_thr, groups, thrs, flags, capacity)
            except RuntimeError as e:       # (TORCH_CHECK of the binding: the same exception type as the ctypes path)
                raise _lib.ObbOpsError(str(e)) from None
            if rc != _lib.OBB_EWORKSPACE:
                break
            capacity = max(int(info[3]), 2 * capacity)
            nbytes = _lib.lib().obb_nms_workspace_bytes(n, kind, capacity)
            if nbytes > MAX_WORKSPACE_BYTES:
                raise _lib.ObbOpsError(
                    'NMS on %d boxes needs a %.1f GiB pair workspace (capacity %d pairs): more than '
                    'nms_wrapper.MAX_WORKSPACE_BYTES; the scene is almost all mutual overlaps' %
                    (n, nbytes / 2.0 ** 30, capacity))
        return keep, (info[1], info[2], info[3])
    L = _lib.lib()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    info = getattr(_tls, 'info', None)      # (argtypes convert plain ints: no c_void_p objects per call)
    if info is None:
        info = _tls.info = (ctypes.c_int64 * 4)()
    gptr = tptr = None
    if groups is not None or thrs is not None:
        groups, gptr, thrs, tptr = _side_inputs(dev, n, groups, thrs)
    # OBB_NMS_RETAIN: the workspace is this thread's own per (device, stream) and nothing else writes it between calls
    # (_workspace; the asynchronous tickets of nms_enqueue use it too -- the library forgets a workspace's state then)
    flags = (_lib.NMS_STRICT if strict else 0) | _lib.NMS_RETAIN
    # (this function is the per-call overhead of every NMS: no context-manager object unless the
    # device has to be switched)
    cur = torch.cuda.current_device()
    if cur != dev.index:
        torch.cuda.set_device(dev)
    try:
        stream = _lib.current_stream_handle(dev)
        capacity = 0  # library default
        while True:
            ws, nbytes = _workspace(L, dev, stream, n, kind, capacity)
            rc = L.obb_nms_ex_f32(dets_dev.data_ptr(), n, kind, iou_thr, gptr, tptr, flags,
                                  keep.data_ptr(), info, ws.data_ptr(), nbytes, stream)
            if rc != _lib.OBB_EWORKSPACE:
                break
            # a pair capacity the library says suffices; every request at least doubles, so the
            # loop ends (at the latest in _workspace's byte budget)
            capacity = max(int(info[3]), 2 * capacity)
    finally:
        if cur != dev.index:
            torch.cuda.set_device(cur)
    if rc != _lib.OBB_OK:
        _lib.check(rc)
    return keep.narrow(0, 0, info[0]), (info[1], info[2], info[3])


def _run_nms_f64(dets_dev, iou_thr, groups=None, thrs=None, strict=False):
    """float64 horizontal boxes (n, 5) on the device -> int64 keep: ``obb_nms_hbb_f64`` (the reference computes float64
    input in double, nms_cpu.cpp:62-68)."""
    n, dev = dets_dev.size(0), dets_dev.device
    L = _lib.lib()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    info = (ctypes.c_int64 * 4)()
    groups, gptr, thrs, tptr = _side_inputs(dev, n, groups, thrs)
    with _lib.on_device(dev):
        stream = _lib.current_stream_handle(dev)
        capacity = 0
        while True:
            ws, nbytes = _workspace(L, dev, stream, n, _lib.KIND_HBB, capacity)
            rc = L.obb_nms_hbb_f64(dets_dev.data_ptr(), n, float(iou_thr), gptr, tptr, _lib.NMS_STRICT if strict else 0,
                                   keep.data_ptr(), info, ws.data_ptr(), nbytes, stream)
            if rc != _lib.OBB_EWORKSPACE:
                break
            capacity = max(int(info[3]), 2 * capacity)
    _lib.check(rc)
    return keep.narrow(0, 0, info[0])


_warned_downcast = set()


def _warn_downcast(name):
    """The rotated operators compute in fp32 (the reference's GPU ops are float-only, nms_kernel.cu:76; its CPU NMS would
    follow the input's dtype, nms_cpu.cpp:62-68): a float64 input is down-cast, and says so once per operator."""
    if name not in _warned_downcast:
        _warned_downcast.add(name)
        import warnings
        warnings.warn('%s: float64 input is computed in float32 on the GPU (only the horizontal `nms` / `soft_nms` / '
                      '`bbox_overlaps` have a float64 path)' % name, UserWarning, stacklevel=3)


def _as_tensor(x):
    if x is None or isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x))


def _nms_generic(dets, iou_thr, device_id, cols, kind, name, groups=None, thrs=None, strict=False):
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
    elif kind == _lib.KIND_HBB and dets_th.dtype == torch.float64:
        # the reference dispatches on the scalar type (nms_cpu.cpp:62-68): double boxes are decided in double
        _lib.require_gpu()
        dev = dets_th.device if dets_th.is_cuda else torch.device('cuda', torch.cuda.current_device() if device_id is None else device_id)
        d = dets_th.detach().to(device=dev).contiguous()
        inds = _run_nms_f64(d, float(iou_thr), _as_tensor(groups), _as_tensor(thrs), strict)
        if not is_numpy:
            inds = inds.to(dets_th.device)
    else:
        if dets_th.dtype == torch.float64:
            _warn_downcast(name)
        d = _lib.to_device_f32(dets_th, device_id)
        inds, _ = _run_nms(d, kind, float(iou_thr), _as_tensor(groups), _as_tensor(thrs), strict)
        if not is_numpy:
            inds = inds.to(dets_th.device)  # keep lives where the input lives
    if is_numpy:
        inds = inds.cpu().numpy()
        return dets[inds, :], inds
    return _take_rows(dets, inds), inds


def _take_rows(dets, inds):
    """dets[inds, :] -- fp32 rows on the GPU through the library's row gather (torch's advanced indexing
    costs 20-40 us for 1e5 rows of 24 bytes, a third of a 100k-box call)."""
    # (the raw kernel has no autograd history: rows that must stay differentiable w.r.t. dets, as the reference's
    # dets[inds, :] are, take torch's indexing whatever their number)
    differentiable = torch.is_grad_enabled() and dets.requires_grad
    if (dets.is_cuda and dets.dtype == torch.float32 and inds.numel() >= 4096 and inds.device == dets.device
            and not differentiable):   # (below 4096 rows the ctypes call costs more)
        src = dets.contiguous()
        out = src.new_empty((inds.numel(), src.size(1)))
        with _lib.on_device(src.device):
            _lib.check(_lib.lib().obb_gather_rows_f32(src.data_ptr(), src.size(0), src.size(1), inds.data_ptr(),
                                                      inds.numel(), src.size(1), out.data_ptr(),
                                                      _lib.current_stream_handle(src.device)))
        return out
    return dets[inds, :]


def rnms(dets, iou_thr, device_id=None, groups=None, thrs=None, strict=False):
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
        strict (bool): suppress on ``>`` (the reference's CUDA kernel,
            nms_kernel.cu:61) instead of ``>=``.

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
    return _nms_generic(dets, iou_thr, device_id, 6, _lib.KIND_RBOX, 'rnms', groups, thrs, strict)


def poly_nms(dets, iou_thr, device_id=None, groups=None, thrs=None, strict=False):
    """Greedy NMS on convex quadrilaterals: dets (n, 9) <x1,y1,..,x4,y4,score>
    (the 8-point polygons the reference writes for the DOTA merge,
    mmdet/datasets/dota.py:197-235, merged by the external mergebypoly)."""
    return _nms_generic(dets, iou_thr, device_id, 9, _lib.KIND_POLY, 'poly_nms', groups, thrs, strict)


# the names the reference's own (disabled) call sites use
thetaobb_nms = rnms      # mmdet/core/post_processing/rbbox_nms.py:97
pointobb_nms = poly_nms


def nms(dets, iou_thr, device_id=None, groups=None, thrs=None, strict=False):
    """Horizontal-box NMS with the reference's signature (nms_wrapper.py:7):
    dets (n, 5) <x1, y1, x2, y2, score> -> (dets[inds], inds).

    The pairs are decided with the reference's own arithmetic -- areas
    ``(x2-x1+1)*(y2-y1+1)``, intersection widths ``min(x2)-max(x1)+1``,
    ``inter / (a_i + a_j - inter) >= iou_thr`` (nms_cpu.cpp:19,47-56) -- in a dedicated
    narrow phase of the HIP kernels (OBB_KIND_HBB), not with the rotated clip.
    ``iou_thr <= 0`` is served like the reference serves it: ``ovr >= iou_thr`` (nms_cpu.cpp:56) then holds for
    DISJOINT boxes too, so every later box of the array (of its group) falls to the first one unless its ``ovr`` is NaN
    (two zero-area boxes) -- a dedicated all-pairs kernel, pinned by tests/golden/hbb_nms_nonpos.npz.  Per-box
    thresholds (``thrs``, an extension without a reference counterpart) must be positive under ``>=``: host arrays are
    checked, a HIP tensor is its owner's responsibility (checking it would cost a kernel and a device-to-host
    synchronisation before the NMS is even queued); with a non-positive entry the result is the broad phase's -- only
    overlapping pairs count -- not nms_cpu.cpp's.

    Example (nms_wrapper.py:24-34):
        >>> dets = np.array([[49.1, 32.4, 51.0, 35.9, 0.9],
        >>>                  [49.3, 32.9, 51.0, 35.3, 0.9],
        >>>                  [49.2, 31.8, 51.0, 35.4, 0.5],
        >>>                  [35.1, 11.5, 39.1, 15.7, 0.5],
        >>>                  [35.6, 11.8, 39.3, 14.2, 0.5],
        >>>                  [35.3, 11.5, 39.9, 14.5, 0.4],
        >>>                  [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)
        >>> suppressed, inds = nms(dets, 0.7)
        >>> assert len(inds) == len(suppressed) == 3
    """
    if not strict and thrs is not None:
        # the same rule box by box: a per-box threshold <= 0 would have to suppress every later box of its group,
        # disjoint ones included (nms_cpu.cpp:56), which the kernels' broad phase never even tests
        # (checked for HOST thresholds only -- a list, an array, a CPU tensor: a device tensor would cost a kernel and a
        # device-to-host synchronisation before the NMS is even queued; its owner vouches for it)
        t = _as_tensor(thrs)
        if not t.is_cuda and t.numel() and not bool((t > 0).all()):
            raise ValueError('nms: every per-box threshold must be positive under the `>=` rule')
    return _nms_generic(dets, iou_thr, device_id, 5, _lib.KIND_HBB, 'nms', groups, thrs, strict)


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3, device_id=None):
    """Soft-NMS with the reference's signature (nms_wrapper.py:63-118), on the GPU.

    The reference runs it on the CPU only (:64); here one HIP workgroup replays
    ``soft_nms_cpu_kernel`` (nms_cpu.cpp:71-210) including its in-place array permutation,
    so equal scores resolve identically.

    Arguments:
        dets (torch.Tensor or np.ndarray): (n, 5) bboxes with scores.
        iou_thr (float): IoU threshold of the linear method.
        method (str): 'linear' or 'gaussian'.
        sigma (float): gaussian decay parameter.
        min_score (float): boxes whose decayed score falls below it are dropped.

    Returns:
        tuple: new dets (k, 5) with decayed scores, in selection order, and their original
        indices (k,) -- same container type, dtype and device as the input.

    Example (nms_wrapper.py:80-89):
        >>> dets = np.array([[4., 3., 5., 3., 0.9], [4., 3., 5., 4., 0.9], [3., 1., 3., 1., 0.5],
        >>>                  [3., 1., 3., 1., 0.5], [3., 1., 3., 1., 0.4], [3., 1., 3., 1., 0.0]],
        >>>                 dtype=np.float32)
        >>> new_dets, inds = soft_nms(dets, 0.7, sigma=0.5)
        >>> assert len(inds) == len(new_dets) == 3
    """
    if isinstance(dets, torch.Tensor):
        is_tensor = True
        dets_th = dets.detach()
    elif isinstance(dets, np.ndarray):
        is_tensor = False
        dets_th = torch.from_numpy(dets)
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(
            type(dets)))
    method_codes = {'linear': 1, 'gaussian': 2}
    if method not in method_codes:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    if dets_th.dim() != 2 or dets_th.size(1) != 5:
        raise ValueError('soft_nms expects dets of shape (n, 5), got {}'.format(tuple(dets_th.shape)))
    n = dets_th.size(0)
    if n == 0:
        new_dets = dets_th.new_zeros((0, 5), dtype=torch.float32)
        inds = torch.zeros(0, dtype=torch.long)
    elif dets_th.dtype == torch.float64 and n <= SOFT_NMS_F64_MAX:
        # the reference dispatches on the scalar type (nms_cpu.cpp:212-215): float64 rows are decayed in double
        _lib.require_gpu()
        dev = dets_th.device if dets_th.is_cuda else torch.device('cuda', torch.cuda.current_device() if device_id is None else device_id)
        d = dets_th.to(device=dev).contiguous()
        out = torch.empty((n, 5), dtype=torch.float64, device=dev)
        oinds = torch.empty(n, dtype=torch.int64, device=dev)
        count = ctypes.c_int64(0)
        with _OnDevice(dev):
            _lib.check(_lib.lib().obb_soft_nms_f64(d.data_ptr(), n, float(iou_thr), method_codes[method], float(sigma), float(min_score),
                                                   out.data_ptr(), oinds.data_ptr(), ctypes.byref(count), _lib.current_stream_handle(dev)))
        new_dets, inds = out[:count.value], oinds[:count.value]
    else:
        if dets_th.dtype == torch.float64:
            _warn_downcast('soft_nms (more than %d float64 rows)' % SOFT_NMS_F64_MAX)
        d = _lib.to_device_f32(dets_th, device_id)
        dev = d.device
        L = _lib.lib()
        out = torch.empty((n, 5), dtype=torch.float32, device=dev)
        oinds = torch.empty(n, dtype=torch.int64, device=dev)
        ws = torch.empty(L.obb_soft_nms_workspace_bytes(n), dtype=torch.uint8, device=dev)
        count = ctypes.c_int64(0)
        with _OnDevice(dev):
            rc = L.obb_soft_nms_f32(d.data_ptr(), n, float(iou_thr), method_codes[method], float(sigma),
                                    float(min_score), out.data_ptr(), oinds.data_ptr(),
                                    ctypes.byref(count), ws.data_ptr(), ws.numel(),
                                    _lib.current_stream_handle(dev))
        _lib.check(rc)
        new_dets, inds = out[:count.value], oinds[:count.value]
    if is_tensor:
        return (new_dets.to(device=dets.device, dtype=dets.dtype),
                inds.to(device=dets.device, dtype=torch.long))
    return new_dets.cpu().numpy().astype(dets.dtype), inds.cpu().numpy().astype(np.int64)


def soft_nms_grouped(dets, seg_sizes, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """Soft-NMS of several independent groups in ONE launch (``obb_soft_nms_grouped_f32``): the rows of ``dets`` (n, 5)
    are grouped contiguously, ``seg_sizes`` (list of ints or int tensor) their sizes.  Group g gets exactly the result
    of ``soft_nms(dets[group g], ...)`` -- the per-class calls of ``multiclass_nms_with_index`` (rbbox_nms.py:29-49) --
    but the groups run concurrently, one workgroup each, with one host read for all the counts.

    Returns:
        list of (new_dets (k_g, 5), inds (k_g,) relative to the group's first row), one pair per group.
    """
    method_codes = {'linear': 1, 'gaussian': 2}
    if method not in method_codes:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    if not (isinstance(dets, torch.Tensor) and dets.dim() == 2 and dets.size(1) == 5):
        raise ValueError('soft_nms_grouped expects a (n, 5) tensor')
    d = _lib.to_device_f32(dets.detach())
    dev = d.device
    sizes = torch.as_tensor(seg_sizes, dtype=torch.int64).reshape(-1)
    G, n = sizes.numel(), d.size(0)
    seg = torch.zeros(G + 1, dtype=torch.int64)
    seg[1:] = torch.cumsum(sizes.cpu(), 0)
    assert int(seg[-1]) == n, 'segment sizes must add up to the number of rows'
    if G == 0:
        return []
    L = _lib.lib()
    seg_dev = seg.to(dev)
    out = torch.empty((n, 5), dtype=torch.float32, device=dev)
    oinds = torch.empty(n, dtype=torch.int64, device=dev)
    counts = torch.empty(G, dtype=torch.int64, device=dev)
    ws = torch.empty(max(int(L.obb_soft_nms_workspace_bytes(n)), 256), dtype=torch.uint8, device=dev)
    with _OnDevice(dev):
        _lib.check(L.obb_soft_nms_grouped_max_f32(d.data_ptr(), n, seg_dev.data_ptr(), G, int(sizes.max()) if G else 0, float(iou_thr), method_codes[method],
                                              float(sigma), float(min_score), out.data_ptr(), oinds.data_ptr(),
                                              counts.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_handle(dev)))
    res = []
    for g, (o, k) in enumerate(zip(seg[:-1].tolist(), counts.tolist())):      # (the one synchronisation)
        res.append((out[o:o + k].to(device=dets.device, dtype=dets.dtype), oinds[o:o + k].to(dets.device)))
    return res


# ---------------------------------------------------------------------------------------------
# Asynchronous form: queue several NMS calls, collect them with ONE synchronisation.
# A detector calls NMS once per FPN level (rpn_head.py:92) and once per class
# (rbbox_nms.py:43); each synchronous call costs a host round trip of ~40 us on top of the
# kernels.  `nms_enqueue` returns a ticket, `nms_collect` turns tickets into keep tensors.
# ---------------------------------------------------------------------------------------------
_KINDS = {'nms': (_lib.KIND_HBB, 5), 'rnms': (_lib.KIND_RBOX, 6), 'poly_nms': (_lib.KIND_POLY, 9)}


class NmsTicket(object):
    __slots__ = ('dets', 'kind', 'iou_thr', 'groups', 'thrs', 'strict', 'keep', 'record', 'nbytes', 'event')


def nms_enqueue(dets, iou_thr, op='nms', groups=None, thrs=None, strict=False):
    """Queue ``op`` ('nms' | 'rnms' | 'poly_nms') on ``dets`` (fp32 HIP tensor) on the current
    stream without waiting for it (obb_nms_async_f32).  The calls of one stream may share the
    thread's workspace: they execute in order."""
    kind, cols = _KINDS[op]
    if not (isinstance(dets, torch.Tensor) and dets.is_cuda):
        raise TypeError('nms_enqueue takes a HIP tensor')
    if dets.dim() != 2 or dets.size(1) != cols:
        raise ValueError('{} expects dets of shape (n, {}), got {}'.format(op, cols, tuple(dets.shape)))
    d = dets.detach().to(dtype=torch.float32).contiguous()
    n, dev = d.size(0), d.device
    t = NmsTicket()
    t.dets, t.kind, t.iou_thr, t.strict = d, kind, float(iou_thr), bool(strict)
    t.groups, gptr, t.thrs, tptr = _side_inputs(dev, n, _as_tensor(groups), _as_tensor(thrs))
    t.keep = torch.empty(n, dtype=torch.int64, device=dev)
    t.record = torch.empty(_lib.NMS_RESULT_BYTES // 8, dtype=torch.int64, device=dev)
    L = _lib.lib()
    with _OnDevice(dev):
        stream = _lib.current_stream_handle(dev)
        ws, t.nbytes = _workspace(L, dev, stream, n, kind, 0)
        _lib.check(L.obb_nms_async_f32(d.data_ptr(), n, kind, t.iou_thr, gptr, tptr,
                                       _lib.NMS_STRICT if strict else 0, t.keep.data_ptr(),
                                       t.record.data_ptr(), ws.data_ptr(), t.nbytes, stream))
        # the kernels run on the stream that is current NOW; whoever collects may be on another one
        t.event = torch.cuda.Event()
        t.event.record(torch.cuda.current_stream(dev))
    return t


def nms_collect(tickets):
    """Keep tensors (ascending int64 indices) of queued calls: one device-to-host copy of all
    result records.  A call whose pair lists overflowed the default workspace is repeated
    synchronously with the capacity the library asks for."""
    if not tickets:
        return []
    L = _lib.lib()
    # one copy per device (tickets of one device: one stacked copy).  The copy runs on the stream that is
    # current HERE, which first waits for the event every ticket recorded on ITS stream at enqueue time
    # (tickets queued under `with torch.cuda.stream(s)` -- one side stream per FPN level -- and
    # collected outside it would otherwise be read before their kernels finished); the keep tensors
    # handed out below are ordered behind the same wait.
    by_dev = {}
    for i, t in enumerate(tickets):
        by_dev.setdefault(t.record.device, []).append(i)
    recs = [None] * len(tickets)
    for dev, idx in by_dev.items():
        cur = torch.cuda.current_stream(dev)
        for i in idx:
            cur.wait_event(tickets[i].event)
        with _lib.on_device(dev):
            host = torch.stack([tickets[i].record for i in idx]).cpu().numpy()
        for row, i in zip(host, idx):
            recs[i] = row
    info = (ctypes.c_int64 * 4)()
    out = []
    for t, rec in zip(tickets, recs):
        rc = L.obb_nms_async_result(rec.ctypes.data, t.dets.size(0), t.nbytes, info)
        if rc == _lib.OBB_EWORKSPACE:
            keep, _ = _run_nms(t.dets, t.kind, t.iou_thr, t.groups, t.thrs, t.strict)
        else:
            _lib.check(rc)
            keep = t.keep[:info[0]]
        out.append(keep)
    return out
