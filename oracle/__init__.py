"""ctypes front-end of the CPU oracle (oracle/obb_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  Nothing under aidet_amd/ may import this module.

Parity status: the rotated arithmetic is "parity unpinned" (the reference has no
rotated ops, see obb_oracle.c header); control flow, theta=0 values and RoIAlign
tap rules are pinned by tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libobb_oracle.so')


def build(force=False):
    """Compile libobb_oracle.so with gcc (seconds) when it is missing.  (Staleness
    is make's job: __graft_entry__.build() runs `make -C oracle`; comparing mtimes
    here would race when several ranks import the module on a fresh snapshot.)"""
    if force or not os.path.exists(_SO):
        subprocess.check_call(['make', '-C', _HERE, 'libobb_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.obbo_sincos_f32.argtypes = [ctypes.c_float, _f32p, _f32p]
        L.obbo_sincos_f32.restype = None
        L.obbo_rbox_iou_f32.argtypes = [_f32p, _f32p, ctypes.c_int]
        L.obbo_rbox_iou_f32.restype = ctypes.c_float
        L.obbo_overlaps_f32.argtypes = [_f32p, ctypes.c_long, _f32p, ctypes.c_long,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.obbo_overlaps_f32.restype = ctypes.c_int
        L.obbo_overlaps_f64.argtypes = [_f64p, ctypes.c_long, _f64p, ctypes.c_long,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, _f64p]
        L.obbo_overlaps_f64.restype = ctypes.c_int
        L.obbo_score_key.argtypes = [ctypes.c_float]
        L.obbo_score_key.restype = ctypes.c_uint32
        L.obbo_score_order.argtypes = [_f32p, ctypes.c_long, ctypes.c_long, _i64p]
        L.obbo_score_order.restype = ctypes.c_int
        L.obbo_nms_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, ctypes.c_float, _i64p]
        L.obbo_nms_f32.restype = ctypes.c_long
        L.obbo_nms_grouped_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, ctypes.c_float,
                                           ctypes.POINTER(ctypes.c_int32), _f32p, _i64p]
        L.obbo_nms_grouped_f32.restype = ctypes.c_long
        L.obbo_hbb_nms_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_float, _i64p]
        L.obbo_hbb_nms_f32.restype = ctypes.c_long
        L.obbo_hbb_overlaps_f32.argtypes = [_f32p, ctypes.c_long, _f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, _f32p]
        L.obbo_hbb_overlaps_f32.restype = ctypes.c_int
        L.obbo_nms_ex_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_int, ctypes.c_float,
                                      ctypes.POINTER(ctypes.c_int32), _f32p, ctypes.c_int, _i64p]
        L.obbo_nms_ex_f32.restype = ctypes.c_long
        L.obbo_soft_nms_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_float, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, _f32p]
        L.obbo_soft_nms_f32.restype = ctypes.c_long
        L.obbo_polyiou_f64.argtypes = [_f64p, _f64p, ctypes.c_long, ctypes.c_int, ctypes.c_int, _f64p]
        L.obbo_polyiou_f64.restype = ctypes.c_int
        L.obbo_max_threads.restype = ctypes.c_int
        L.obbo_overlaps_mt_f32.argtypes = [_f32p, ctypes.c_long, _f32p, ctypes.c_long,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.obbo_overlaps_mt_f32.restype = ctypes.c_int
        L.obbo_nms_mt_f32.argtypes = [_f32p, ctypes.c_long, ctypes.c_float, ctypes.c_int, _i64p]
        L.obbo_nms_mt_f32.restype = ctypes.c_long
        rroi = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                _f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                ctypes.c_float, ctypes.c_int, ctypes.c_float, _f32p]
        L.obbo_rroi_align_forward_f32.argtypes = rroi
        L.obbo_rroi_align_forward_f32.restype = ctypes.c_int
        L.obbo_rroi_align_backward_f32.argtypes = rroi
        L.obbo_rroi_align_backward_f32.restype = ctypes.c_int
        L.obbo_rroi_align_forward_mt_f32.argtypes = rroi[:-1] + [ctypes.c_int, _f32p]
        L.obbo_rroi_align_forward_mt_f32.restype = ctypes.c_int
        L.obbo_roi_align_forward_f32.argtypes = [
            _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            _f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
            ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.obbo_roi_align_forward_f32.restype = ctypes.c_int
        L.obbo_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def sincos(theta):
    s = ctypes.c_float()
    c = ctypes.c_float()
    lib().obbo_sincos_f32(ctypes.c_float(float(np.float32(theta))), ctypes.byref(s), ctypes.byref(c))
    return np.float32(s.value), np.float32(c.value)


def _mode(mode):
    assert mode in ('iou', 'iof')
    return 0 if mode == 'iou' else 1


def overlaps(a, b, mode='iou', is_aligned=False, kind='rbox'):
    """fp32 spec overlaps. kind 'rbox': rows (cx,cy,w,h,theta); 'poly': 8 coords.

    Return shapes follow geometry.py:51-88 of the reference: (m,n), or (m,) when
    aligned.
    """
    cols = 5 if kind == 'rbox' else 8
    a = _f32(a).reshape(-1, cols)
    b = _f32(b).reshape(-1, cols)
    m, n = a.shape[0], b.shape[0]
    out = np.zeros((m,) if is_aligned else (m, n), dtype=np.float32)
    if is_aligned:
        assert m == n
    if m * n == 0:
        return out
    rc = lib().obbo_overlaps_f32(_p(a), m, _p(b), n, 0 if kind == 'rbox' else 1,
                                 _mode(mode), 1 if is_aligned else 0, _p(out))
    assert rc == 0
    return out


def bbox_overlaps(a, b, mode='iou', is_aligned=False):
    """Restatement of the reference's horizontal bbox_overlaps (geometry.py:47-88), fp32:
    (m,4),(n,4) xyxy -> (m,n) or (m,)."""
    a = _f32(a).reshape(-1, 4)
    b = _f32(b).reshape(-1, 4)
    m, n = a.shape[0], b.shape[0]
    out = np.zeros((m,) if is_aligned else (m, n), dtype=np.float32)
    if m * n == 0:
        return out
    rc = lib().obbo_hbb_overlaps_f32(_p(a), m, _p(b), n, _mode(mode), 1 if is_aligned else 0, _p(out))
    assert rc == 0
    return out


def overlaps_f64(a, b, mode='iou', is_aligned=False, kind='rbox'):
    """fp64 truth: libm trig, no prefilter, vertex-list Sutherland-Hodgman."""
    cols = 5 if kind == 'rbox' else 8
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, cols)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, cols)
    m, n = a.shape[0], b.shape[0]
    out = np.zeros((m,) if is_aligned else (m, n), dtype=np.float64)
    if m * n == 0:
        return out
    rc = lib().obbo_overlaps_f64(_p(a, _f64p), m, _p(b, _f64p), n,
                                 0 if kind == 'rbox' else 1, _mode(mode),
                                 1 if is_aligned else 0, _p(out, _f64p))
    assert rc == 0
    return out


def score_order(scores):
    s = _f32(scores).reshape(-1)
    order = np.zeros(s.shape[0], dtype=np.int64)
    lib().obbo_score_order(_p(s), s.shape[0], 1, _p(order, _i64p))
    return order


def _nms(d, kind, iou_thr, groups, thrs, strict=False):
    n = d.shape[0]
    keep = np.zeros(n, dtype=np.int64)
    g = t = None
    gp = ctypes.POINTER(ctypes.c_int32)()
    tp = _f32p()
    if groups is not None:
        g = np.ascontiguousarray(groups, dtype=np.int32).reshape(n)
        gp = g.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    if thrs is not None:
        t = _f32(thrs).reshape(n)
        tp = _p(t)
    k = lib().obbo_nms_ex_f32(_p(d), n, kind, ctypes.c_float(iou_thr), gp, tp, 1 if strict else 0,
                              _p(keep, _i64p))
    assert k >= 0
    return keep[:k].copy()


def rnms(dets, iou_thr, groups=None, thrs=None, strict=False):
    """dets (n,6) cx,cy,w,h,theta,score -> ascending int64 keep indices.
    groups (n,) int: only boxes of the same group interact; thrs (n,) per-box thr;
    strict: suppress on `>` (nms_kernel.cu:61) instead of `>=` (nms_cpu.cpp:56)."""
    return _nms(_f32(dets).reshape(-1, 6), 0, iou_thr, groups, thrs, strict)


def poly_nms(dets, iou_thr, groups=None, thrs=None, strict=False):
    """dets (n,9) x1,y1..x4,y4,score -> ascending int64 keep indices."""
    return _nms(_f32(dets).reshape(-1, 9), 1, iou_thr, groups, thrs, strict)


def hbb_nms(dets, iou_thr, groups=None, strict=False):
    """Restatement of nms_cpu.cpp:5-60 for (n,5) x1,y1,x2,y2,score (fp32), with the
    reference's own pair arithmetic (+1 widths, :19,47-55); optional groups / strict `>`."""
    d = _f32(dets).reshape(-1, 5)
    if groups is None and not strict:
        keep = np.zeros(d.shape[0], dtype=np.int64)
        k = lib().obbo_hbb_nms_f32(_p(d), d.shape[0], ctypes.c_float(iou_thr), _p(keep, _i64p))
        assert k >= 0
        return keep[:k].copy()
    return _nms(d, 2, iou_thr, groups, None, strict)


def soft_nms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """Restatement of soft_nms_cpu_kernel (nms_cpu.cpp:71-210) through the wrapper's
    argument rules (nms_wrapper.py:63-118).  Returns (new_dets (k,5), inds (k,) int64)."""
    codes = {'naive': 0, 'linear': 1, 'gaussian': 2}
    if method not in codes:
        raise ValueError('invalid method for SoftNMS: {}'.format(method))
    d = _f32(dets).reshape(-1, 5)
    out = np.zeros((d.shape[0], 6), dtype=np.float32)
    k = lib().obbo_soft_nms_f32(_p(d), d.shape[0], ctypes.c_float(iou_thr), codes[method],
                                ctypes.c_float(sigma), ctypes.c_float(min_score), _p(out))
    assert k >= 0
    return out[:k, :5].copy(), out[:k, 5].astype(np.int64)


def hbb_nms_f64(dets, iou_thr):
    """nms_cpu_kernel<double> (nms_cpu.cpp:5-68): (n, 5) float64 rows -> ascending int64 keep indices; the threshold is a
    float in the reference's signature and is promoted in the comparison."""
    d = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 5)
    keep = np.zeros(d.shape[0], dtype=np.int64)
    L = lib()
    L.obbo_hbb_nms_f64.restype = ctypes.c_long
    k = L.obbo_hbb_nms_f64(_p(d, _f64p), ctypes.c_long(d.shape[0]), ctypes.c_float(iou_thr), _p(keep, ctypes.POINTER(ctypes.c_int64)))
    assert k >= 0
    return keep[:k].copy()


def soft_nms_f64(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """soft_nms_cpu_kernel<double> (nms_cpu.cpp:71-215) -> (new_dets (k, 5) float64, inds (k,) int64)."""
    codes = {'naive': 0, 'linear': 1, 'gaussian': 2}
    d = np.ascontiguousarray(dets, dtype=np.float64).reshape(-1, 5)
    out = np.zeros((d.shape[0], 6), dtype=np.float64)
    L = lib()
    L.obbo_soft_nms_f64.restype = ctypes.c_long
    k = L.obbo_soft_nms_f64(_p(d, _f64p), ctypes.c_long(d.shape[0]), ctypes.c_float(iou_thr), codes[method], ctypes.c_float(sigma),
                            ctypes.c_float(min_score), _p(out, _f64p))
    assert k >= 0
    return out[:k, :5].copy(), out[:k, 5].astype(np.int64)


def bbox_overlaps_f64(a, b, mode='iou', is_aligned=False):
    """geometry.py:47-88 on float64 boxes (numpy restatement of the same expressions)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    if is_aligned:
        lt, rb = np.maximum(a[:, :2], b[:, :2]), np.minimum(a[:, 2:], b[:, 2:])
        wh = np.clip(rb - lt + 1, 0, None)
        ov = wh[:, 0] * wh[:, 1]
        a1 = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
        if mode == 'iou':
            a2 = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
            return ov / (a1 + a2 - ov)
        return ov / a1
    lt, rb = np.maximum(a[:, None, :2], b[None, :, :2]), np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt + 1, 0, None)
    ov = wh[..., 0] * wh[..., 1]
    a1 = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    if mode == 'iou':
        a2 = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return ov / (a1[:, None] + a2[None, :] - ov)
    return ov / a1[:, None]


def polyiou_f64(a, b, kind='rbox', recentre=False):
    """Aligned-pair IoU by the DOTA devkit's triangle-fan algorithm, fp64 (third,
    independent formulation; see obb_oracle.c)."""
    cols = 5 if kind == 'rbox' else 8
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, cols)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, cols)
    assert a.shape == b.shape
    out = np.zeros(a.shape[0], dtype=np.float64)
    if a.shape[0]:
        rc = lib().obbo_polyiou_f64(_p(a, _f64p), _p(b, _f64p), a.shape[0], 0 if kind == 'rbox' else 1,
                                    1 if recentre else 0, _p(out, _f64p))
        assert rc == 0
    return out


def max_threads():
    return int(lib().obbo_max_threads())


def overlaps_mt(a, b, threads, mode='iou'):
    """obbo_overlaps_f32 with rows dealt to `threads` OpenMP threads (same values)."""
    a = _f32(a).reshape(-1, 5)
    b = _f32(b).reshape(-1, 5)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if out.size:
        rc = lib().obbo_overlaps_mt_f32(_p(a), a.shape[0], _p(b), b.shape[0], 0, _mode(mode), int(threads), _p(out))
        assert rc == 0
    return out


def rnms_mt(dets, iou_thr, threads):
    """rnms with the suppression loop of each kept box dealt to threads (same keep)."""
    d = _f32(dets).reshape(-1, 6)
    keep = np.zeros(d.shape[0], dtype=np.int64)
    k = lib().obbo_nms_mt_f32(_p(d), d.shape[0], ctypes.c_float(iou_thr), int(threads), _p(keep, _i64p))
    assert k >= 0
    return keep[:k].copy()


def rroi_align_forward(feat, rois, out_size, spatial_scale, sample_num=0, min_size=1.0, threads=1):
    feat = _f32(feat)
    rois = _f32(rois).reshape(-1, 6)
    B, C, H, W = feat.shape
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), dtype=np.float32)
    if K and threads > 1:
        rc = lib().obbo_rroi_align_forward_mt_f32(_p(feat), B, C, H, W, _p(rois), K, ph, pw,
                                                  ctypes.c_float(spatial_scale), sample_num,
                                                  ctypes.c_float(min_size), int(threads), _p(out))
        assert rc == 0
    elif K:
        rc = lib().obbo_rroi_align_forward_f32(_p(feat), B, C, H, W, _p(rois), K, ph, pw,
                                               ctypes.c_float(spatial_scale), sample_num,
                                               ctypes.c_float(min_size), _p(out))
        assert rc == 0
    return out


def rroi_align_backward(grad_out, rois, feat_shape, spatial_scale, sample_num=0, min_size=1.0):
    g = _f32(grad_out)
    rois = _f32(rois).reshape(-1, 6)
    B, C, H, W = feat_shape
    K, _, ph, pw = g.shape
    gin = np.zeros((B, C, H, W), dtype=np.float32)
    if K:
        rc = lib().obbo_rroi_align_backward_f32(_p(g), B, C, H, W, _p(rois), K, ph, pw,
                                                ctypes.c_float(spatial_scale), sample_num,
                                                ctypes.c_float(min_size), _p(gin))
        assert rc == 0
    return gin


def roi_align_forward(feat, rois, out_size, spatial_scale, sample_num=0, version=2, aligned=False):
    """Horizontal RoIAlign V1/V2 restatement (anchor for RRoIAlign at theta=0)."""
    feat = _f32(feat)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat.shape
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    K = rois.shape[0]
    out = np.zeros((K, C, ph, pw), dtype=np.float32)
    if K:
        rc = lib().obbo_roi_align_forward_f32(_p(feat), B, C, H, W, _p(rois), K, ph, pw,
                                              ctypes.c_float(spatial_scale), sample_num,
                                              version, 1 if aligned else 0, _p(out))
        assert rc == 0
    return out


# ---------------------------------------------------------------------------
# synthetic workloads shared by tests and bench (SURVEY.md 8d / BASELINE.md 4)
# ---------------------------------------------------------------------------
def synth_boxes(n, canvas, seed):
    """(n,6) fp32: cx,cy ~ U(0,canvas); w,h = exp(U(ln8,ln128)); theta ~ U(-pi/2,0);
    score = (perm(n)+0.5)/n (all distinct)."""
    rng = np.random.default_rng(seed)
    cx = rng.uniform(0, canvas, n)
    cy = rng.uniform(0, canvas, n)
    w = np.exp(rng.uniform(np.log(8.0), np.log(128.0), n))
    h = np.exp(rng.uniform(np.log(8.0), np.log(128.0), n))
    th = rng.uniform(-np.pi / 2, 0.0, n)
    sc = (rng.permutation(n) + 0.5) / n
    return np.stack([cx, cy, w, h, th, sc], axis=1).astype(np.float32)


def rbox_to_poly(boxes):
    """(n,5+) theta-OBB -> (n,8) corners in cv2.boxPoints order
    (transforms.py:45-55): p0 = c-u+v, p1 = c-u-v, p2 = c+u-v, p3 = c+u+v."""
    b = np.asarray(boxes, dtype=np.float64)
    c, s = np.cos(b[:, 4]), np.sin(b[:, 4])
    ux, uy = 0.5 * b[:, 2] * c, 0.5 * b[:, 2] * s
    vx, vy = -0.5 * b[:, 3] * s, 0.5 * b[:, 3] * c
    cx, cy = b[:, 0], b[:, 1]
    return np.stack([cx - ux + vx, cy - uy + vy, cx - ux - vx, cy - uy - vy,
                     cx + ux - vx, cy + uy - vy, cx + ux + vx, cy + uy + vy], axis=1)


def synth_quads(n, canvas, seed, jitter=0.25):
    """(n,9) fp32 random CONVEX quadrilaterals + score: rectangles whose corners are
    moved by up to `jitter` of the shorter side (stays convex for jitter < 0.3), in
    random vertex order direction (CW / CCW)."""
    b = synth_boxes(n, canvas, seed)
    p = rbox_to_poly(b).reshape(n, 4, 2)
    rng = np.random.default_rng(seed + 7919)
    side = np.minimum(b[:, 2], b[:, 3])[:, None, None]
    p = p + rng.uniform(-jitter, jitter, p.shape) * side
    flip = rng.random(n) < 0.5
    p[flip] = p[flip][:, ::-1]
    return np.concatenate([p.reshape(n, 8), b[:, 5:6]], 1).astype(np.float32)
