"""GPU tests of the round-5 additions (run on the MI355X box: `pytest -m gpu`)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star: RoI-Align floats within 1e-4 of the CPU restatement


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device (no CPU path exists)'
    return torch.device('cuda:0')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def _rois(k, canvas, seed, batch=1):
    r = oracle.synth_boxes(k, canvas, seed)[:, :5]
    b = (np.arange(k) % batch).astype(np.float32)[:, None]
    return np.concatenate([b, r], 1).astype(np.float32)


def _hrois(k, canvas, seed, batch=1):
    rng = np.random.default_rng(seed)
    x1 = rng.uniform(0, canvas * 0.9, k); y1 = rng.uniform(0, canvas * 0.9, k)
    w = np.exp(rng.uniform(np.log(4), np.log(canvas / 3), k)); h = np.exp(rng.uniform(np.log(4), np.log(canvas / 3), k))
    b = (np.arange(k) % batch)
    return np.stack([b, x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


@pytest.fixture
def nhwc_nq():
    """forces the slab width of the channels-last forward (lanes per tap); default restored afterwards"""
    from aidet_amd import _lib

    def force(nq):
        _lib.set_tuning(_lib.TUNE_RROI_NHWC_NQ, nq)
    yield force
    _lib.set_tuning(_lib.TUNE_RROI_NHWC_NQ, -1)


# ---- channels-last RoI-Align forward ---------------------------------------------------------------
@pytest.mark.parametrize('nq', [-1, 8, 16, 32, 64])
@pytest.mark.parametrize('C,H,W,K,out,sn', [(16, 64, 64, 40, 7, 2), (40, 32, 32, 10, 14, 1), (256, 64, 64, 64, 7, 2),
                                            (12, 33, 47, 25, (5, 9), 0), (260, 40, 24, 30, 7, 2), (64, 50, 50, 20, (8, 8), 3)])
def test_nhwc_forward_is_bit_equal_to_the_nchw_kernel_and_meets_the_oracle(C, H, W, K, out, sn, nq, nhwc_nq):
    """A channels-last map is pooled in place (obb_roi_align_forward_nhwc_f32): bit-equal to the NCHW kernel on the
    same values (same expressions, same order of the sums), <= 1e-4 from the oracle; every slab width; C that is not
    a multiple of the slab (260 = 256 + 4), several tiles (14 x 14 bins), adaptive grids (sample_num 0: several table
    rounds), an even bin count."""
    from aidet_amd import RRoIAlign, roi_layers
    rng = np.random.default_rng(C + K)
    feat = rng.standard_normal((2, C, H, W)).astype(np.float32)
    rois = _rois(K, 4 * W, 5, batch=2)
    rois[0, 0] = 7            # a batch index outside the map: zeros
    rois[1, 3:5] = 0.0        # a degenerate RoI (min_size clamps it)
    layer = RRoIAlign(out, 0.25, sample_num=sn)
    ft = t(feat)
    nchw = layer(ft, t(rois))
    nhwc_nq(nq)
    fl = ft.contiguous(memory_format=torch.channels_last)
    assert roi_layers.is_nhwc(fl)
    got = layer(fl, t(rois))
    assert got.is_contiguous() and got.shape == nchw.shape
    assert torch.equal(got, nchw)
    ref = oracle.rroi_align_forward(feat, rois, out, 0.25, sn)
    assert np.abs(got.cpu().numpy() - ref).max() <= TOL


@pytest.mark.parametrize('aligned', [False, True])
def test_nhwc_forward_horizontal_kinds(aligned, nhwc_nq):
    """RoIAlign V1 / V2-aligned rows (roi_align_kernel.cu:64-117, roi_align_kernel_v2.cu:62-128) on a channels-last map:
    bit-equal to the NCHW kernel, which is bit-equal to the reference's expressions (test_gpu_parity)."""
    from aidet_amd import RoIAlign
    rng = np.random.default_rng(11)
    feat = rng.standard_normal((2, 32, 48, 56)).astype(np.float32)
    rois = _hrois(60, 200, 3, batch=2)
    rois[5, 3] = rois[5, 1] - 3.0      # x2 < x1: V1 clamps the size at 0 -> 0 / 0, V2 samples backwards
    for sn in (2, 0):
        layer = RoIAlign(7, 0.25, sample_num=sn, aligned=aligned)
        a = layer(t(feat), t(rois))
        b = layer(t(feat).contiguous(memory_format=torch.channels_last), t(rois))
        assert torch.equal(torch.nan_to_num(a, nan=123.0), torch.nan_to_num(b, nan=123.0))
        ref = oracle.roi_align_forward(feat, rois, 7, 0.25, sn, version=2 if aligned else 1, aligned=aligned)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isnan(ref), np.isnan(b.cpu().numpy()))
        assert np.abs(b.cpu().numpy()[ok] - ref[ok]).max() <= TOL


def test_nhwc_forward_cfg3_and_many_rois(nhwc_nq):
    """BASELINE.json configs[2] (512 rotated RoIs, 256 x 256 x 256 map, 7 x 7) on a channels-last map: <= 1e-4 from the
    oracle, bit-equal to the NCHW kernel; 2000 RoIs (the workspace route of the Python layer, 8 waves per block): the same."""
    from aidet_amd import RRoIAlign, _lib
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = _rois(512, 1024, 5)
    layer = RRoIAlign(7, 0.25, sample_num=2)
    ref = oracle.rroi_align_forward(feat, rois, 7, 0.25, 2)
    fl = t(feat).contiguous(memory_format=torch.channels_last)
    got = layer(fl, t(rois))
    assert np.abs(got.cpu().numpy() - ref).max() <= TOL
    assert torch.equal(got, layer(t(feat), t(rois)))
    many = t(_rois(2000, 1024, 8))
    want = layer(t(feat), many)
    for nw in (-1, 4, 8):
        _lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, nw)
        try:
            assert torch.equal(layer(fl, many), want)
        finally:
            _lib.set_tuning(_lib.TUNE_RROI_NHWC_WAVES, -1)


def test_nhwc_pyramid_extractor_and_fallbacks(nhwc_nq):
    """SingleRoIExtractor (single_level.py:11-107) over a channels-last pyramid = the NCHW pyramid, in one launch; a
    pyramid with one NCHW level, a half-precision map and C % 4 != 0 go through the NCHW copy as before."""
    from aidet_amd import SingleRoIExtractor, RotatedSingleRoIExtractor, RRoIAlign, roi_layers
    rng = np.random.default_rng(3)
    feats = [t(rng.standard_normal((2, 64, 128 >> i, 160 >> i)).astype(np.float32)) for i in range(4)]
    cl = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    hro = t(_hrois(300, 600, 4, batch=2))
    ex = SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=2), 64, [4, 8, 16, 32])
    with torch.no_grad():
        a = ex(feats, hro)
        assert torch.equal(ex(cl, hro), a)
        assert torch.equal(ex([cl[0], feats[1], cl[2], cl[3]], hro), a)            # mixed: copied to NCHW
    rro = t(_rois(300, 600, 6, batch=2))
    rex = RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=7, sample_num=2), 64, [4, 8, 16, 32])
    with torch.no_grad():
        assert torch.equal(rex(cl, rro), rex(feats, rro))
    layer = RRoIAlign(7, 0.25, sample_num=2)
    odd = t(rng.standard_normal((1, 6, 20, 20)).astype(np.float32))
    assert not roi_layers.is_nhwc(odd.contiguous(memory_format=torch.channels_last))
    r = t(_rois(9, 80, 2))
    assert torch.equal(layer(odd.contiguous(memory_format=torch.channels_last), r), layer(odd, r))


def test_nhwc_forward_c_abi_rejects_what_it_cannot_take():
    from aidet_amd import _lib
    L = _lib.lib()
    f = torch.zeros((1, 8, 8, 6), device=dev())
    r = torch.zeros((1, 6), device=dev())
    o = torch.zeros((1, 6, 7, 7), device=dev())
    rc = L.obb_rroi_align_forward_nhwc_f32(f.data_ptr(), 1, 6, 8, 8, r.data_ptr(), 1, 7, 7, 1.0, 2, 1.0, o.data_ptr(), None)
    assert rc == _lib.OBB_EINVAL and b'multiple of 4' in L.obb_last_error()
    rc = L.obb_rroi_align_forward_nhwc_f32(f.data_ptr() + 4, 1, 8, 8, 6, r.data_ptr(), 1, 7, 7, 1.0, 2, 1.0, o.data_ptr(), None)
    assert rc == _lib.OBB_EINVAL and b'aligned' in L.obb_last_error()
    assert L.obb_rroi_align_forward_nhwc_f32(None, 1, 8, 8, 8, None, 0, 7, 7, 1.0, 2, 1.0, None, None) == _lib.OBB_OK   # K = 0


# ---- channels-last RoI-Align backward ------------------------------------------------------------------------------
@pytest.mark.parametrize('C,H,W,K,out,sn,batch', [(24, 40, 40, 50, 7, 2, 2), (256, 64, 64, 64, 7, 2, 1), (8, 33, 47, 25, (5, 9), 0, 2),
                                                  (260, 24, 40, 30, 7, 2, 1), (64, 20, 20, 300, 7, 2, 1), (16, 70, 70, 12, 14, 2, 1),
                                                  (16, 24, 24, 700, 7, 2, 1)])
def test_nhwc_backward_is_bit_equal_to_the_nchw_gather_and_meets_the_oracle(C, H, W, K, out, sn, batch):
    """obb_roi_backward_gather_nhwc_f32 writes the gradient of a channels-last map in place: bit-equal to the NCHW gather
    (same summation order, fused multiply-adds) where every RoI is planned, <= 1e-4 max|g| from the oracle; small RoIs
    (long tap lists per pixel: several rounds of 64 entries), many RoIs on one tile (several batches; 700 RoIs: the tile's list
    is summed in two passes, the first stored and read back), C = 260
    (two channel slabs), a RoI outside the batch, a tile overhanging the map (33 x 47), 14 x 14 bins (196 > 64: those
    RoIs go through the atomics kernel)."""
    from aidet_amd import RRoIAlign, roi_layers
    rng = np.random.default_rng(C + K)
    feat = rng.standard_normal((batch, C, H, W)).astype(np.float32)
    rois = _rois(K, 4 * W, 9, batch=batch)
    rois[0, 0] = batch + 3
    if K >= 600:
        rois[:, 3:5] = np.clip(rois[:, 3:5], 40.0, 90.0)       # 700 RoIs of 10 .. 22 map pixels on a 24 x 24 map: a tile lists > 512
    elif K >= 300:
        rois[:, 3:5] = np.minimum(rois[:, 3:5], 14.0)          # tiny RoIs: ~50-90 taps per pixel
    layer = RRoIAlign(out, 0.25, sample_num=sn)
    g = rng.standard_normal((K, C) + ((out, out) if isinstance(out, int) else out)).astype(np.float32)
    fa = t(feat).requires_grad_(True)
    layer(fa, t(rois)).backward(t(g))
    fb = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert roi_layers.is_nhwc(fb)
    layer(fb, t(rois)).backward(t(g))
    assert fb.grad.is_contiguous(memory_format=torch.channels_last) and fb.grad.shape == fa.grad.shape
    ref = oracle.rroi_align_backward(g, rois, feat.shape, 0.25, sn)
    tol = 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(fb.grad.cpu().numpy() - ref).max() <= tol
    nb = out * out if isinstance(out, int) else out[0] * out[1]
    if nb <= 64 and sn > 0:          # every RoI planned on both sides: the same sums in the same order
        assert torch.equal(fb.grad.contiguous(), fa.grad)
    again = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    layer(again, t(rois)).backward(t(g))
    if nb <= 64 and sn > 0:
        assert torch.equal(again.grad, fb.grad)                  # reproducible run to run


@pytest.mark.parametrize('aligned', [False, True])
def test_nhwc_backward_horizontal_kinds(aligned):
    from aidet_amd import RoIAlign
    rng = np.random.default_rng(17)
    feat = rng.standard_normal((2, 32, 48, 56)).astype(np.float32)
    rois = _hrois(60, 200, 3, batch=2)
    layer = RoIAlign(7, 0.25, sample_num=2, aligned=aligned)
    g = rng.standard_normal((60, 32, 7, 7)).astype(np.float32)
    fa = t(feat).requires_grad_(True)
    layer(fa, t(rois)).backward(t(g))
    fb = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    layer(fb, t(rois)).backward(t(g))
    assert torch.equal(fb.grad.contiguous(), fa.grad)
    # the op is linear in the map: <grad_feat, feat> == <g, forward(feat)> (the forward is pinned against the oracle's
    # restatement of roi_align_kernel.cu / roi_align_kernel_v2.cu above)
    out = layer(fb.detach(), t(rois)).double()
    lhs = float((fb.grad.double() * fb.detach().double()).sum()); rhs = float((t(g).double() * out).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(rhs))


def test_nhwc_backward_cfg3_and_large_rois():
    """configs[2] shape; and RoIs larger than the plan holds (> 2047 footprint pixels): the atomics tail."""
    from aidet_amd import RRoIAlign
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = _rois(512, 1024, 5)
    rois[:6, 3:5] = 400.0              # 100 x 100 map pixels each
    layer = RRoIAlign(7, 0.25, sample_num=2)
    g = rng.standard_normal((512, 256, 7, 7)).astype(np.float32)
    fb = t(feat).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    layer(fb, t(rois)).backward(t(g))
    ref = oracle.rroi_align_backward(g, rois, feat.shape, 0.25, 2)
    assert np.abs(fb.grad.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    fa = t(feat).requires_grad_(True)
    layer(fa, t(rois)).backward(t(g))
    assert np.abs(fb.grad.cpu().numpy() - fa.grad.cpu().numpy()).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_nhwc_backward_c_abi_edges():
    from aidet_amd import _lib
    L = _lib.lib()
    assert L.obb_roi_backward_nhwc_workspace_bytes(512, 256, 7, 7) >= 512 * 256 * 49 * 4
    gm = torch.full((1, 8, 8, 8), 7.0, device=dev())
    rc = L.obb_roi_backward_gather_nhwc_f32(None, 1, 8, 8, 8, None, 0, 0, 7, 7, 1.0, 2, 1.0, gm.data_ptr(), None, 0, None)
    torch.cuda.synchronize()
    assert rc == _lib.OBB_OK and float(gm.abs().max()) == 0.0           # K = 0: the map is overwritten with zeros
    g = torch.zeros((1, 6, 7, 7), device=dev()); r = torch.zeros((1, 6), device=dev())
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev())
    rc = L.obb_roi_backward_gather_nhwc_f32(g.data_ptr(), 1, 6, 8, 8, r.data_ptr(), 1, 0, 7, 7, 1.0, 2, 1.0, gm.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == _lib.OBB_EINVAL and b'multiple of 4' in L.obb_last_error()
    g = torch.zeros((1, 8, 7, 7), device=dev())
    rc = L.obb_roi_backward_gather_nhwc_f32(g.data_ptr(), 1, 8, 8, 8, r.data_ptr(), 1, 0, 7, 7, 1.0, 2, 1.0, gm.data_ptr(), ws.data_ptr(), 16, None)
    assert rc == _lib.OBB_EWORKSPACE


# ---- nms with a non-positive threshold (VERDICT r4 "weak" 2) ------------------------------------------------------------
def test_hbb_nms_nonpositive_threshold_matches_reference(golden):
    """`nms(dets, iou_thr <= 0)` returns what nms_cpu.cpp:56 returns (fixtures produced by running the reference): float32
    and float64 rows, numpy and HIP containers, the queued form, groups (= the oracle's loop per group), a NaN threshold
    (nothing is suppressed)."""
    from aidet_amd import nms, nms_wrapper
    g = golden('hbb_nms_nonpos.npz')
    for tag in ('a', 'b', 'inv', 'zero', 'zerotop'):
        for ti, thr in enumerate(g['thrs'].tolist()):
            for dn in ('f32', 'f64'):
                d, keep = g['%s_%s_dets' % (tag, dn)], g['%s_%s_keep%d' % (tag, dn, ti)]
                kd, ki = nms(d, thr)
                assert isinstance(ki, np.ndarray) and ki.dtype == np.int64 and np.array_equal(ki, keep), (tag, dn, thr)
                assert np.array_equal(kd, d[keep])
                assert np.array_equal(nms(t(d), thr)[1].cpu().numpy(), keep), (tag, dn, thr)
            d32 = g['%s_f32_dets' % tag]
            got = nms_wrapper.nms_collect([nms_wrapper.nms_enqueue(t(d32), thr)])[0]
            assert np.array_equal(got.cpu().numpy(), g['%s_f32_keep%d' % (tag, ti)])
    d = g['inv_f32_dets']
    grp = (np.arange(len(d)) % 7).astype(np.int32)
    for thr in (0.0, -0.5):
        assert np.array_equal(nms(t(d), thr, groups=grp)[1].cpu().numpy(), oracle.hbb_nms(d, thr, groups=grp))
    assert np.array_equal(nms(t(d), float('nan'))[1].cpu().numpy(), np.arange(len(d)))
    assert np.array_equal(oracle.hbb_nms(d, float('nan')), np.arange(len(d)))


def test_hbb_nms_nonpositive_threshold_c_abi():
    """the same through the bare C ABI: obb_nms_f32 no longer refuses the call"""
    from aidet_amd import _lib
    L = _lib.lib()
    d = np.array([[0, 0, 10, 10, 0.9], [100, 100, 110, 110, 0.8], [50, 50, 49, 60, 0.95], [7, 7, 6, 9, 0.1]], np.float32)
    dd = t(d)
    keep = torch.empty(4, dtype=torch.int64, device=dev())
    ws = torch.empty(L.obb_nms_workspace_bytes(4, _lib.KIND_HBB, 0), dtype=torch.uint8, device=dev())
    info = (ctypes.c_int64 * 4)()
    rc = L.obb_nms_f32(dd.data_ptr(), 4, _lib.KIND_HBB, 0.0, keep.data_ptr(), info, ws.data_ptr(), ws.numel(), None)
    assert rc == _lib.OBB_OK, L.obb_last_error()
    assert keep[:info[0]].cpu().tolist() == oracle.hbb_nms(d, 0.0).tolist()


# ---- small completions (VERDICT r4 "missing" 3, 4) ---------------------------------------------------------------------
def test_c_host_runs_the_gpu_through_the_abi(tmp_path):
    """tests/c/abi_gpu.c: a C99 program (HIP runtime C API for memory and the stream, nothing else) drives obb_nms_f32,
    obb_overlaps_f32 and obb_rroi_align_forward_nhwc_f32 on the GPU and compares with values embedded from the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'aidet_amd', 'lib')
    exe = str(tmp_path / 'abi_gpu')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(root, 'include'), '-I', '/opt/rocm/include',
                           os.path.join(root, 'tests', 'c', 'abi_gpu.c'), '-o', exe, '-L', libdir, '-lobbops', '-L', '/opt/rocm/lib',
                           '-lamdhip64', '-lm', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert 'c host on the gpu ok' in out.stdout


def test_aug_test_of_the_detector():
    """``ThetaOBBRCNN.aug_test`` (rbbox_cnn.py:255-282 with test_mixins.py:37-59,122-157): two views of one tile (the
    tile, and the tile flipped at half the size) -> merged proposals -> boxes decoded in every view, mapped back,
    averaged, class-wise NMS.  Checked here: the plumbing against the same steps composed by hand from the tested parts."""
    import torch.nn.functional as F
    from aidet_amd.bbox_transforms import bbox2roi, bbox_mapping, delta2bbox
    from aidet_amd.merge_augs import merge_aug_bboxes, merge_aug_proposals
    from aidet_amd.models.theta_obb_rcnn import build_detector, demo_inputs
    from aidet_amd.post_processing import multiclass_nms
    model = build_detector(dev())
    img, meta = demo_inputs(dev(), size=512)
    small = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False).flip(-1)
    metas = [meta, [dict(img_shape=(256, 256, 3), ori_shape=(512, 512, 3), pad_shape=(256, 256, 3), scale_factor=0.5, flip=True)]]
    # (the conv trunk is stock PyTorch / MIOpen, whose first call of a shape may pick another solver than the next: the
    #  three evaluations below share ONE set of feature maps, so that what is compared is this repo's plumbing)
    with torch.no_grad():
        cache = {id(im): model.extract_feat(im) for im in (img, small)}
    model.extract_feat = lambda im: cache[id(im)]
    model.aug_test([img, small], metas, rescale=True, with_rotated=True)      # (warm-up: the FC heads' GEMM solvers settle)
    res = model.aug_test([img, small], metas, rescale=True)
    assert len(res) == 15 and all(r.ndim == 2 and r.shape[1] == 5 for r in res)
    both = model.aug_test([img, small], metas, rescale=True, with_rotated=True)
    assert len(both) == 2 and len(both[1]) == 15
    assert all(r.ndim == 2 and r.shape[1] == 6 for r in both[1]), [r.shape for r in both[1]]
    assert all(a.shape == b.shape and np.allclose(a, b, atol=1e-4) for a, b in zip(both[0], res))
    # the same steps by hand
    with torch.no_grad():
        feats = [model.extract_feat(im) for im in (img, small)]
        props = []
        for x, m in zip(feats, metas):
            cs, bp = model.rpn_head(x)
            props.append(model.rpn_head.get_bboxes(cs, bp, m, model.test_cfg['rpn'], fused=True)[0])
        merged = merge_aug_proposals(props, [m[0] for m in metas], model.test_cfg['rpn'])
        ab, asc = [], []
        for x, m in zip(feats, metas):
            rois = bbox2roi([bbox_mapping(merged[:, :4], m[0]['img_shape'], m[0]['scale_factor'], m[0]['flip'])])
            cls, reg = model.bbox_head(model.bbox_roi_extractor(x[:4], rois))
            ab.append(delta2bbox(rois[:, 1:], reg, (0., 0., 0., 0.), model.bbox_target_stds, m[0]['img_shape']))
            asc.append(F.softmax(cls, dim=1))
        mb, msc = merge_aug_bboxes(ab, asc, metas, None)
        det, lab = multiclass_nms(mb, msc, 0.05, dict(type='nms', iou_thr=0.5), 1000)
    for c in range(15):
        want = det[lab == c].cpu().numpy()
        assert res[c].shape == want.shape and np.allclose(res[c], want, atol=1e-4), c
    assert sum(len(r) for r in res) == det.size(0) > 0



# ---- Soft-NMS on one wave (VERDICT r4 "next" 4) --------------------------------------------------------------------------
@pytest.mark.parametrize('method', ['linear', 'gaussian'])
def test_soft_nms_one_wave_kernel_equals_the_host_loop_and_the_workgroup_kernel(method):
    """soft_nms_wave_kernel (arrays of up to 64 boxes by default; OBB_TUNE_SOFT_NMS_SHAPE = 4 forces it up to 512: NB = 1, 2,
    4, 8 boxes per lane): index- and bit-equal to the oracle's loop (nms_cpu.cpp:71-210) and to the workgroup kernel
    (shape 3) at every size class, on dense and sparse scenes, with score ties (the kernel gives up, the array kernel
    replays the permutation) and drops in the last rounds."""
    from aidet_amd import _lib
    from aidet_amd.nms_wrapper import soft_nms
    rng = np.random.default_rng(11)
    for n in (1, 2, 63, 64, 65, 128, 129, 200, 256, 257, 400, 512):
        for canvas, ties in ((120, False), (900, False), (300, True)):
            x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
            sc = (rng.permutation(n) + 0.5) / n
            if ties:
                sc = np.round(sc * 8) / 8
            hb = np.stack([x1, y1, x1 + rng.uniform(8, 120, n), y1 + rng.uniform(8, 120, n), sc], 1).astype(np.float32)
            want_d, want_i = oracle.soft_nms(hb, 0.3, method, 0.5, 0.05)
            for shape in (0, 3, 4):
                try:
                    _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, shape)
                    d, i = soft_nms(t(hb), 0.3, method=method, sigma=0.5, min_score=0.05)
                    path = int(_lib.lib().obb_soft_nms_last_path())
                finally:
                    _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, -1)
                assert np.array_equal(i.cpu().numpy(), want_i) and np.array_equal(d.cpu().numpy(), want_d), (n, canvas, ties, shape)
                assert path in ((3,) if ties and n > 8 else (2, 3)), (path, n, ties)


def test_soft_nms_grouped_on_one_wave_per_class():
    """`soft_nms_grouped` with groups of up to 512 boxes: a wave per class, results == the oracle's loop per class."""
    from aidet_amd import nms_wrapper
    rng = np.random.default_rng(4)
    sizes = [0, 1, 70, 300, 512, 129, 0, 64]
    rows, segs = [], [0]
    for m in sizes:
        x1, y1 = rng.uniform(0, 400, m), rng.uniform(0, 400, m)
        rows.append(np.stack([x1, y1, x1 + rng.uniform(8, 100, m), y1 + rng.uniform(8, 100, m), (rng.permutation(m) + 0.5) / max(m, 1)], 1))
        segs.append(segs[-1] + m)
    d = np.concatenate(rows, 0).astype(np.float32)
    res = nms_wrapper.soft_nms_grouped(t(d), sizes, 0.3, method='gaussian', sigma=0.5, min_score=0.05)
    assert len(res) == len(sizes)
    for g, (dd, ii) in enumerate(res):
        part = d[segs[g]:segs[g + 1]]
        wd, wi = oracle.soft_nms(part, 0.3, 'gaussian', 0.5, 0.05) if len(part) else (np.zeros((0, 5), np.float32), np.zeros(0, np.int64))
        assert np.array_equal(ii.cpu().numpy(), wi) and np.array_equal(dd.cpu().numpy(), wd), g


# ---- resolve of large calls (VERDICT r4 "next" 5) -------------------------------------------------------------------------
def test_resolve_paths_of_a_large_call_agree():
    """n >= 2^18 (no fused compaction): four workgroups per edge shard, the first sweep on cached source words, the register
    tail -- every combination returns the keep list of the plain sweeps on a scene with dense clusters and 3000-box chains;
    the same chains alone (one workgroup resolves them: 2999 edges) meet the oracle with the tail on and off."""
    from aidet_amd import _lib, rnms
    n = 280000
    d = oracle.synth_boxes(n, 9000, 11)
    L, m = 3000, 60000
    k = np.arange(m)
    run, pos = k // L, k % L
    d[:m, 0] = run * 400.0 + 20000.0
    d[:m, 1] = pos * 5.0
    d[:m, 2] = 16.0; d[:m, 3] = 16.0; d[:m, 4] = 0.0
    d[:m, 5] = 0.99 - pos / (2.0 * L) - run * 1e-7
    dt = t(d)
    grp = t(np.random.default_rng(3).integers(0, 7, n).astype(np.int32))
    try:
        for groups in (None, grp):
            ref = None
            for mode, parts in ((2, 1), (1, 4), (0, 4), (3, 8), (1, 1)):
                _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)
                _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, parts)
                keep = rnms(dt, 0.3, groups=groups)[1]
                ref = keep if ref is None else ref
                assert torch.equal(keep, ref), (mode, parts)
            assert ref.numel() > 0
        chain = d[:L].copy()
        want = oracle.rnms(chain, 0.3)
        for mode in (0, 1, 2, 3):
            _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, mode)
            assert np.array_equal(rnms(t(chain), 0.3)[1].cpu().numpy(), want), mode
    finally:
        _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, -1)
        _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_PARTS, -1)



def test_a_stalled_resolve_is_an_error_not_a_short_keep_list():
    """OBB_ESTALLED end to end (the resolve's workgroups wait for each other and must be resident together; one that reaches
    its bound publishes rounds = -1): with the test hook (OBB_TUNE_NMS_RESOLVE_TAIL bit 2) every route that resolves with
    nms_resolve_kernel raises instead of returning a keep list, and the next call without the hook is served as usual."""
    from aidet_amd import _lib, rnms
    for n, canvas in ((5000, 2500), (50000, 12000), (300000, 40000)):
        d = t(oracle.synth_boxes(n, canvas, 3))
        want = rnms(d, 0.5)[1]
        try:
            _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, 1 | 4)
            with pytest.raises(_lib.ObbOpsError, match='converg'):
                rnms(d, 0.5)
        finally:
            _lib.set_tuning(_lib.TUNE_NMS_RESOLVE_TAIL, -1)
        assert torch.equal(rnms(d, 0.5)[1], want)
