"""GPU tests of the round-4 additions (run on the MI355X box: `pytest -m gpu`)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device (no CPU path exists)'
    return torch.device('cuda:0')


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def _rpn_select(L, _lib, cls, reg, anchors, A, sig, nms_pre, ws):
    n = len(cls)
    hs = (ctypes.c_int * n)(*[c.size(1) for c in cls]); wd = (ctypes.c_int * n)(*[c.size(2) for c in cls])
    N = sum(min(nms_pre, c.size(1) * c.size(2) * A) for c in cls)
    dets = torch.empty((N, 5), device=dev()); lv = torch.empty(N, dtype=torch.int32, device=dev())
    rows = ctypes.c_int64(0)
    _lib.check(L.obb_rpn_select_decode_ws_f32(
        (ctypes.c_void_p * n)(*[c.data_ptr() for c in cls]), (ctypes.c_void_p * n)(*[r.data_ptr() for r in reg]),
        (ctypes.c_void_p * n)(*[a.data_ptr() for a in anchors]), hs, wd, n, A, 1 if sig else 0, nms_pre,
        (ctypes.c_float * 4)(0, 0, 0, 0), (ctypes.c_float * 4)(1, 1, 1, 1), 16 / 1000.0, 0, 0,
        dets.data_ptr(), lv.data_ptr(), ctypes.byref(rows), None if ws is None else ws.data_ptr(),
        0 if ws is None else ws.numel(), _lib.current_stream_handle(dev())))
    assert rows.value == N
    return dets, lv


def test_rpn_level_barrier_timeout_falls_back_to_one_workgroup():
    """the barrier among a level's workgroups is bounded.  OBB_TUNE_RPN_BARRIER_TICKS = 0 makes
    every wait time out at once: the level's `fail` word goes up, all its workgroups leave, and
    rpn_select_decode_kernel selects the level alone -- the result is the workspace-less path's, bit for bit, and the
    call after it (default bound) is the multi-workgroup result again."""
    from aidet_amd import _lib
    from aidet_amd.anchors import AnchorGenerator
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    A = 3
    sizes = [(256, 256), (128, 128), (64, 64)]
    cls = [torch.randn((A, h, w), generator=g).to(dev()) for (h, w) in sizes]
    reg = [torch.randn((4 * A, c.size(1), c.size(2)), generator=g).to(dev()) * 0.1 for c in cls]
    anchors = [AnchorGenerator(4 << i, [8], [0.5, 1.0, 2.0]).grid_anchors(c.shape[-2:], 4 << i, device=dev()) for i, c in enumerate(cls)]
    n = len(cls)
    hs = (ctypes.c_int * n)(*[c.size(1) for c in cls]); wd = (ctypes.c_int * n)(*[c.size(2) for c in cls])
    ws = torch.empty(int(L.obb_rpn_select_workspace_bytes(hs, wd, n, A, 2000)), dtype=torch.uint8, device=dev())
    plain = _rpn_select(L, _lib, cls, reg, anchors, A, True, 2000, None)
    grouped = _rpn_select(L, _lib, cls, reg, anchors, A, True, 2000, ws)
    assert torch.equal(plain[0], grouped[0]) and torch.equal(plain[1], grouped[1])      # (random logits: no ties at the cut)
    try:
        _lib.set_tuning(_lib.TUNE_RPN_BARRIER_TICKS, 0)
        timed_out = _rpn_select(L, _lib, cls, reg, anchors, A, True, 2000, ws)
        torch.cuda.synchronize()
    finally:
        _lib.set_tuning(_lib.TUNE_RPN_BARRIER_TICKS, -1)
    assert torch.equal(plain[0], timed_out[0]) and torch.equal(plain[1], timed_out[1])
    again = _rpn_select(L, _lib, cls, reg, anchors, A, True, 2000, ws)
    assert torch.equal(plain[0], again[0])


# ---- fused phases of the tiled pass (obb_tiled_local_f32 / obb_tiled_merge_f32) ------------------------------------
def _tiled_args(images=2, image_size=2048, objects=1500, classes=5, seed=5):
    import workloads
    scene = workloads.tiled_scene(images=images, image_size=image_size, objects_per_image=objects, num_classes=classes, seed=seed)
    idx = np.arange(len(scene['dets']), dtype=np.int64)
    return [t(scene['dets']), t(scene['label']), t(scene['patch']), t(idx), t(scene['patch_xy']), t(scene['patch_image'])], scene


@pytest.mark.parametrize('thrs', [None, [0.3, 0.5, 0.1, 0.4, 0.2]])
def test_fused_tiled_phases_equal_the_step_by_step_pass_and_the_oracle(thrs):
    """One pass of configs[4] on one rank through the two fused entry points (groups and per-class thresholds derived
    inside the NMS's prep kernel, the keep count read on the device) == the step-by-step pass (patch groups kernel ->
    rnms -> rows kernel -> split kernel -> rnms -> gather) == the same pipeline on the oracle's NMS."""
    from aidet_amd import dota_merge
    import mp_gpu_cases as cases
    args, scene = _tiled_args()
    marks = []
    fk, fkeep, frows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs, mark=marks.append)
    assert marks == ['local', 'gather', 'merge']
    try:
        dota_merge.FUSED = False
        sk, skeep, srows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs)
    finally:
        dota_merge.FUSED = True
    assert torch.equal(frows, srows) and torch.equal(fkeep, skeep) and torch.equal(fk, sk)
    ok, okeep, orows = dota_merge.tiled_inference_merge(*[a.cpu() for a in args], 5, 0.5, 0.3, thrs, nms_fn=cases.oracle_nms)
    assert np.array_equal(fk.cpu().numpy(), ok.numpy()) and np.array_equal(fkeep.cpu().numpy(), okeep.numpy())
    assert 0 < fk.size(0) < frows.size(0) < args[0].size(0)
    # no detections at all
    empty = [a[:0] for a in args[:4]] + args[4:]
    ek, ekeep, erows = dota_merge.tiled_inference_merge(*empty, 5, 0.5, 0.3, thrs)
    assert ek.shape == (0, 8) and ekeep.numel() == 0 and erows.size(0) == 0


def test_fused_tiled_phases_repeat_a_phase_whose_pair_lists_overflow():
    """3000 copies of one box in one patch and class: 4.5e6 overlapping pairs exceed the default pair capacity of the
    per-patch NMS (the device-side verdict turns K negative, the phase is repeated with the capacity it names); the
    merged NMS of a scene whose duplicates survive (local threshold 1.5: nothing is suppressed) overflows in turn."""
    from aidet_amd import dota_merge
    rng = np.random.default_rng(0)
    n = 3000
    d = np.tile(np.array([[500.0, 400.0, 60.0, 30.0, -0.4, 0.0]], np.float32), (n, 1))
    d[:, 5] = (rng.permutation(n) + 0.5) / n
    base = [t(d), t(np.zeros(n, np.int64)), t(np.zeros(n, np.int64)), t(np.arange(n, dtype=np.int64)),
            t(np.zeros((1, 2), np.float32)), t(np.zeros(1, np.int64))]
    dota_merge._phase_capacity.clear()
    k, keep, rows = dota_merge.tiled_inference_merge(*base, 3, 0.5, 0.3, None)
    assert k.size(0) == 1 and rows.size(0) == 1 and float(k[0, 5]) == float(d[:, 5].max())
    assert (0, n.bit_length()) in dota_merge._phase_capacity            # the local phase overflowed and was repeated
    dota_merge._phase_capacity.clear()
    k, keep, rows = dota_merge.tiled_inference_merge(*base, 3, 1.5, 0.3, None)
    assert rows.size(0) == n and k.size(0) == 1
    assert (1, n.bit_length()) in dota_merge._phase_capacity            # the merge phase overflowed and was repeated
    dota_merge._phase_capacity.clear()


# ---- float64 fidelity (missing #3) --------------------------------------------------------------------------
def test_float64_nms_is_decided_in_double_like_the_reference(golden, nms_route):
    """`nms` on float64 rows -> obb_nms_hbb_f64: areas / intersection / `>=` in double, the order from the 64-bit score
    key; index-exact with the reference's nms_cpu_kernel<double> (fixtures made by running it) on every route,
    numpy and tensor containers, groups."""
    from aidet_amd import nms
    g = golden('hbb_f64.npz')
    thr = {'between': 0.7, 'close': 1.0}
    for route in ('auto', 'direct', 'sorted', 'listed', 'fused'):
        nms_route(route)
        for tag in ('a', 'b', 'c', 'int', 'fine', 'between', 'close'):
            d = g['nms_%s_dets' % tag]
            if route == 'fused' and len(d) > 1024:
                continue
            th = float(g['nms_%s_thr' % tag]) if tag not in thr else thr[tag]
            kept, inds = nms(d, th)
            assert isinstance(inds, np.ndarray) and kept.dtype == np.float64
            assert np.array_equal(inds, g['nms_%s_keep' % tag]), (route, tag)
    nms_route('auto')
    d = g['nms_b_dets']
    kept, inds = nms(torch.from_numpy(d).to(dev()), 0.3)
    assert inds.is_cuda and kept.dtype == torch.float64 and np.array_equal(inds.cpu().numpy(), g['nms_b_keep'])
    # groups: two copies of the scene in two groups = the scene's keep twice
    two = np.concatenate([d, d], 0)
    grp = np.repeat([3, 9], len(d)).astype(np.int32)
    _, gi = nms(two, 0.3, groups=grp)
    assert np.array_equal(gi, np.concatenate([g['nms_b_keep'], g['nms_b_keep'] + len(d)]))
    # larger scenes against the oracle's double restatement (pinned to the fixtures on the CPU side)
    rng = np.random.default_rng(3)
    for n, canvas in ((20000, 1500), (70000, 3000)):
        x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
        b = np.stack([x1, y1, x1 + np.exp(rng.uniform(2, 5, n)), y1 + np.exp(rng.uniform(2, 5, n)), 0.5 + rng.permutation(n) * 1e-12], 1)
        assert np.array_equal(nms(b, 0.5)[1], oracle.hbb_nms_f64(b, 0.5)), n


def test_float64_bbox_overlaps_and_soft_nms_follow_the_input_dtype(golden):
    from aidet_amd import _lib, bbox_overlaps, soft_nms
    g = golden('hbb_f64.npz')
    b1, b2 = torch.from_numpy(g['iou_b1']), torch.from_numpy(g['iou_b2'])
    for where in (lambda x: x, lambda x: x.to(dev())):
        got = bbox_overlaps(where(b1), where(b2))
        assert got.dtype == torch.float64 and np.array_equal(got.cpu().numpy(), g['iou'])
        assert np.array_equal(bbox_overlaps(where(b1), where(b2), mode='iof').cpu().numpy(), g['iof'])
        assert np.array_equal(bbox_overlaps(where(b1), where(b2[:150]), is_aligned=True).cpu().numpy(), g['iou_aligned'])
    for t in 'abc':
        cfg, m = g['soft_%s_cfg' % t], str(g['soft_%s_method' % t])
        nd, inds = soft_nms(g['soft_%s_dets' % t], cfg[0], method=m, sigma=cfg[1], min_score=cfg[2])
        assert nd.dtype == np.float64 and np.array_equal(inds, g['soft_%s_inds' % t]), t
        assert np.array_equal(nd, g['soft_%s_new_dets' % t]), t      # (gaussian too: glibc's double exp restated on the device, soft_exp)
    # fresh double arrays against the oracle's double loop (libm exp): gaussian weights bit for bit
    rng = np.random.default_rng(31)
    for n, canvas in ((300, 80), (1500, 200), (4000, 400)):
        x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
        d = np.stack([x1, y1, x1 + rng.uniform(4, 60, n), y1 + rng.uniform(4, 60, n), rng.uniform(0, 1, n)], 1)
        for sigma in (0.5, 0.2):
            nd, inds = soft_nms(d, 0.3, method='gaussian', sigma=sigma, min_score=0.05)
            rd, ri = oracle.soft_nms_f64(d, 0.3, 'gaussian', sigma, 0.05)
            assert nd.dtype == np.float64 and np.array_equal(inds, ri) and np.array_equal(nd, rd), (n, sigma)
            # (up to 2048 rows the position-free double kernel serves the call -- distinct scores here --, the register kernel above)
            assert _lib.lib().obb_soft_nms_last_path() == (2 if n <= 2048 else 1), n
        # scores that differ beyond float precision only, shared maxima, a NaN, -0: the double order, the array's permutation
        e = d.copy()
        e[:, 4] = 0.5 + np.arange(n) * 1e-12
        f = d.copy()
        f[::6, 4] = 0.5
        if n > 10:
            f[3, 4] = np.nan
            f[7, 4] = -0.0
        for arr, want_path in ((e, 2 if n <= 2048 else 1), (f, 3 if n <= 2048 else 1)):
            for method in ('linear', 'gaussian'):
                nd, inds = soft_nms(arr, 0.3, method=method, sigma=0.5, min_score=0.05)
                rd, ri = oracle.soft_nms_f64(arr, 0.3, method, 0.5, 0.05)
                assert np.array_equal(inds, ri) and np.array_equal(nd, rd, equal_nan=True), (n, method)
                assert _lib.lib().obb_soft_nms_last_path() == want_path, (n, method)


def test_float64_rotated_input_says_that_it_is_down_cast():
    from aidet_amd import nms_wrapper, rnms
    nms_wrapper._warned_downcast.discard('rnms')
    d = oracle.synth_boxes(300, 200, 2).astype(np.float64)
    with pytest.warns(UserWarning, match='float64 input is computed in float32'):
        kept, inds = rnms(d, 0.5)
    assert kept.dtype == np.float64 and np.array_equal(inds, oracle.rnms(d.astype(np.float32), 0.5))


def test_register_soft_nms_replays_the_array_permutation():
    """The register-resident Soft-NMS (positions tracked, boxes never moved) against the oracle's array loop: sizes
    around every launch shape (2 / 4 / 8 boxes per thread, 16 waves, the global-memory kernel), ties, heavy drops
    (min_score 0.3: many rounds compact), drops in the last rounds, NaN / -0 scores, all three methods."""
    from aidet_amd import _lib, soft_nms
    rng = np.random.default_rng(17)
    _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, 1)      # (float rows default to the LDS-array kernel: it is the faster one)
    try:
        _register_soft_nms_cases(soft_nms, rng)
    finally:
        _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, -1)


def _register_soft_nms_cases(soft_nms, rng):
    for n, canvas, min_score in ((1, 50, 0.01), (2, 10, 0.3), (63, 60, 0.3), (500, 120, 0.3), (513, 200, 0.05), (1024, 200, 0.2),
                                 (1500, 250, 0.05), (2048, 260, 0.3), (2049, 300, 0.1), (6000, 500, 0.05), (9000, 900, 0.05)):
        x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
        d = np.stack([x1, y1, x1 + rng.uniform(4, 60, n), y1 + rng.uniform(4, 60, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        d[::5, 4] = np.float32(0.5)                       # ties: the lowest POSITION of the current array wins
        if n > 10:
            d[3, 4] = np.nan; d[7, 4] = -0.0; d[n // 2, 4] = np.nan
        for method, thr in (('linear', 0.3), ('gaussian', 0.3), ('linear', 0.0)):
            nd, inds = soft_nms(d, thr, method=method, sigma=0.5, min_score=min_score)
            rd, ri = oracle.soft_nms(d, thr, method, 0.5, min_score)
            assert np.array_equal(inds, ri), (n, method, thr)
            assert np.array_equal(nd, rd, equal_nan=True), (n, method)      # (gaussian too: glibc's expf restated on the device)


def test_grouped_soft_nms_runs_the_position_free_kernel_first():
    """`soft_nms_grouped`: a workgroup per group on the kernel without positions, the groups it gives up on (shared maxima, NaN
    scores) replayed by the array kernel's launch behind it -- the same rows as the array kernel alone (OBB_TUNE_SOFT_NMS_REG =
    2) and as the oracle's loop per group, gaussian weights bit for bit; sizes around the launch shapes, empty groups, a
    group of one, groups beyond the LDS state."""
    from aidet_amd import _lib, nms_wrapper
    rng = np.random.default_rng(41)
    for sizes in ([400] * 15, [0, 1, 2, 257, 0, 1024, 63], [2049, 300, 4096, 5], [4500, 100, 2000], [30] * 200):
        groups = []
        for gi, m in enumerate(sizes):
            x1, y1 = rng.uniform(0, 500, m), rng.uniform(0, 500, m)
            g = np.stack([x1, y1, x1 + rng.uniform(6, 90, m), y1 + rng.uniform(6, 90, m), rng.random(m)], 1).astype(np.float32)
            if gi % 3 == 1 and m > 4:
                g[::4, 4] = np.float32(0.5)              # shared maxima: the position-free kernel gives this group up
            if gi % 5 == 2 and m > 4:
                g[2, 4] = np.nan
            groups.append(g)
        rows = t(np.concatenate(groups, 0))
        for method, kw in (('linear', dict(min_score=0.05)), ('gaussian', dict(sigma=0.4, min_score=0.02))):
            got = nms_wrapper.soft_nms_grouped(rows, sizes, 0.3, method=method, **kw)
            try:
                _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, 2)
                alone = nms_wrapper.soft_nms_grouped(rows, sizes, 0.3, method=method, **kw)
            finally:
                _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, -1)
            assert len(got) == len(sizes)
            for g, (d, k), (d2, k2) in zip(groups, got, alone):
                rd, ri = oracle.soft_nms(g, 0.3, method, kw.get('sigma', 0.5), kw['min_score'])
                assert np.array_equal(k.cpu().numpy(), ri), (sizes[:4], method, len(g))
                assert np.array_equal(d.cpu().numpy(), rd, equal_nan=True), (sizes[:4], method, len(g))
                assert torch.equal(k, k2) and torch.equal(d.nan_to_num(-7.0), d2.nan_to_num(-7.0))


def test_fused_compaction_of_the_sorted_route_and_its_fallbacks(nms_route):
    """Sorted route below 2^18 boxes: the keep list is written by the tail of the resolve launch (64 workgroups, n / 64
    words each).  Same list as the separate compaction launch (knob 0) and as the path where every waiting shard gives
    up at once and the last one compacts alone (knob 2); all three bit-exact with the oracle."""
    from aidet_amd import _lib
    from aidet_amd.nms_wrapper import _run_nms
    for n, canvas in ((12000, 1500), (70000, 6000), (100000, 20000)):
        d = oracle.synth_boxes(n, canvas, 21)
        want = oracle.rnms(d, 0.5) if n <= 20000 else None
        dd = t(d)
        got = {}
        for route in ('sorted', 'listed'):
            nms_route(route)
            for knob in (1, 0, 2):
                _lib.set_tuning(_lib.TUNE_NMS_FUSED_COMPACT, knob)
                try:
                    keep, info = _run_nms(dd, _lib.KIND_RBOX, 0.5)
                finally:
                    _lib.set_tuning(_lib.TUNE_NMS_FUSED_COMPACT, -1)
                got[(route, knob)] = keep.cpu().numpy()
        first = got[('sorted', 1)]
        assert all(np.array_equal(first, v) for v in got.values()), n
        if want is not None:
            assert np.array_equal(first, want)
    nms_route('auto')


def test_library_side_capacity_hint_serves_c_abi_callers():
    """the capacity hint lives in the library (per calling thread, kind, size class),
    not in the Python wrapper.  A C-ABI caller that always sizes its workspace with obb_nms_workspace_bytes(n, kind, 0):
    the first call of a scene of mutual duplicates overflows and names a capacity, the NEXT default-sized workspace is
    large enough at once, and a sparse scene of the same size class drops the hint again."""
    from aidet_amd import _lib
    L = _lib.lib()
    n = 3000
    rng = np.random.default_rng(1)
    dense = np.tile(np.array([[500.0, 400.0, 60.0, 30.0, -0.4, 0.0]], np.float32), (n, 1))
    dense[:, 5] = (rng.permutation(n) + 0.5) / n
    sparse = oracle.synth_boxes(n, 20000, 3)
    keep = torch.empty(n, dtype=torch.int64, device=dev())
    info = (ctypes.c_int64 * 4)()
    st = _lib.current_stream_handle(dev())

    def call(d):
        nbytes = int(L.obb_nms_workspace_bytes(n, _lib.KIND_RBOX, 0))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        rc = L.obb_nms_f32(d.data_ptr(), n, _lib.KIND_RBOX, 0.5, keep.data_ptr(), info, ws.data_ptr(), nbytes, st)
        return rc, nbytes, int(info[0]), int(info[3])
    call(t(sparse))                                  # (whatever hint an earlier test left: a sparse call drops it)
    rc0, b0, _, _ = call(t(sparse))
    assert rc0 == 0
    rc1, b1, _, need = call(t(dense))
    assert rc1 == _lib.OBB_EWORKSPACE and b1 == b0 and need > 64 * n
    rc2, b2, k2, _ = call(t(dense))                  # the default workspace now follows the hint
    assert rc2 == 0 and b2 > b1 and k2 == 1
    rc3, b3, k3, _ = call(t(sparse))                 # ... and is given up by a call that does not need it
    assert rc3 == 0 and b3 == b2 and k3 == len(oracle.rnms(sparse, 0.5))
    rc4, b4, _, _ = call(t(sparse))
    assert rc4 == 0 and b4 == b0


def test_channels_last_feature_map_is_copied_with_a_warning_once():
    """a channels-last map the kernels cannot pool in place is converted to NCHW before pooling --
    same values, and it says so.  (Since round 5 channels-last maps with C % 4 == 0 ARE pooled in place, silently:
    tests/test_round5_gpu.py; the copy and its warning remain for the rest -- here C = 30 -- and for
    roi_layers.NHWC_NATIVE = False.)"""
    from aidet_amd import RRoIAlign, roi_layers
    rng = np.random.default_rng(4)
    feat = t(rng.standard_normal((1, 30, 48, 40)).astype(np.float32))
    rois = t(np.concatenate([np.zeros((40, 1), np.float32), oracle.synth_boxes(40, 160, 5)[:, :5]], 1))
    layer = RRoIAlign(7, 0.25, sample_num=2)
    want = layer(feat, rois)
    roi_layers._warned_layout.clear()
    with pytest.warns(UserWarning, match='channels-last layout'):
        got = layer(feat.contiguous(memory_format=torch.channels_last), rois)
    assert torch.equal(got, want)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        layer(feat.contiguous(memory_format=torch.channels_last), rois)      # (once per cause)
        layer(feat, rois)
        feat32 = t(rng.standard_normal((1, 32, 48, 40)).astype(np.float32))
        layer(feat32.contiguous(memory_format=torch.channels_last), rois)    # (pooled in place: nothing to warn about)
    old = roi_layers.NHWC_NATIVE
    try:
        roi_layers.NHWC_NATIVE = False
        roi_layers._warned_layout.clear()
        with pytest.warns(UserWarning, match='channels-last layout'):
            got = layer(feat32.contiguous(memory_format=torch.channels_last), rois)
        assert torch.equal(got, layer(feat32, rois))
    finally:
        roi_layers.NHWC_NATIVE = old


def test_planned_roi_forward_is_bit_equal_to_the_unplanned_one():
    """obb_roi_align_forward_ws_f32 with the per-RoI plan pass (rroi_fwd_plan_kernel: frame, footprint, copy reciprocals
    and sample table once per RoI instead of once per block) forced for every RoI count against the same call without
    it and against the workspace-free entry points: rotated and horizontal kinds, a pyramid, RoIs on every path
    (tiny, large footprints, outside the map, invalid batch index), C not a multiple of 32."""
    from aidet_amd import _lib
    from aidet_amd.roi_layers import RotatedSingleRoIExtractor, SingleRoIExtractor
    rng = np.random.default_rng(8)
    feats = [t(rng.standard_normal((2, 72, 64 >> i, 96 >> i)).astype(np.float32)) for i in range(3)]
    K = 700
    x1, y1 = rng.uniform(-20, 380, K), rng.uniform(-20, 250, K)
    size = np.exp(rng.uniform(np.log(2), np.log(500), (K, 2)))
    rois = np.concatenate([rng.integers(0, 2, (K, 1)), x1[:, None], y1[:, None], x1[:, None] + size[:, :1], y1[:, None] + size[:, 1:]], 1).astype(np.float32)
    rois[5, 0] = 7                                   # invalid batch index: zeros
    rr = np.concatenate([rois[:, :1], oracle.synth_boxes(K, 380, 4)[:, :5]], 1).astype(np.float32)
    rr[:, 3:5] *= rng.choice([0.1, 1.0, 6.0], (K, 1))
    ext = SingleRoIExtractor(dict(type='RoIAlign', out_size=7, sample_num=2), 72, [4, 8, 16])
    rext = RotatedSingleRoIExtractor(dict(type='RRoIAlign', out_size=(5, 7), sample_num=2), 72, [4, 8, 16])
    got = {}
    for knob in (0, 1 << 30):
        _lib.set_tuning(_lib.TUNE_RROI_PLAN, knob)
        try:
            with torch.no_grad():
                got[knob] = (ext(feats, t(rois)), rext(feats, t(rr)))
        finally:
            _lib.set_tuning(_lib.TUNE_RROI_PLAN, -1)
    assert torch.equal(got[0][0], got[1 << 30][0]) and torch.equal(got[0][1], got[1 << 30][1])
    assert float(got[0][0].abs().sum()) > 0 and bool((got[0][0][5] == 0).all())


def test_grid_route_is_exact_and_gives_up_on_crowded_scenes(nms_route):
    """The uniform-grid route (frame -> insert -> probe + clip -> resolve + compaction: four launches): bit-exact with
    the oracle / the sorted route on sparse scenes incl. polygons, horizontal boxes, per-box thresholds, far outliers,
    non-finite rows and boxes that touch many cells; a crowded scene (every cell over capacity) makes it give up -- the
    call is repeated on the sorted route (same result) and the thread skips the grid for the size class afterwards."""
    from aidet_amd import _lib, nms, poly_nms, rnms
    from aidet_amd.nms_wrapper import _run_nms
    L = _lib.lib()
    rng = np.random.default_rng(31)
    # sparse scenes of several sizes: the grid serves them
    for n, canvas in ((300, 2000), (5000, 6000), (20000, 9000), (100000, 20000), (200000, 30000)):
        d = oracle.synth_boxes(n, canvas, 40 + n % 7)
        if n == 5000:
            d[7] = [np.nan, 5, 5, 5, 0, 0.9]; d[8] = [100, 100, 0, 0, 0, 0.8]; d[9] = [3e7, -2e7, 20, 30, -0.3, 0.7]   # NaN, empty, far away
            d[10, 2:5] = [2500.0, 40.0, 0.0]                                           # a long box over ~20 cells
        nms_route('grid')
        got = rnms(t(d), 0.5)[1].cpu().numpy()
        route = L.obb_nms_last_route()
        nms_route('sorted')
        want = rnms(t(d), 0.5)[1].cpu().numpy()
        assert L.obb_nms_last_route() == 2
        assert np.array_equal(got, want), n
        assert route == 4, (n, route)
        if n <= 20000:
            assert np.array_equal(got, oracle.rnms(d, 0.5)), n
    # boxes that touch more cells than an entry list may name go on the BIG list and are tested against every box: a few
    # huge boxes among the small ones (two of them on top of each other), per-box thresholds, groups
    d = oracle.synth_boxes(5000, 6000, 9)
    d[11, 2:5] = [3000.0, 2500.0, -0.7]
    d[12, :5] = d[11, :5] + np.float32([20.0, -15.0, 0.0, 0.0, 0.02])
    d[13, :5] = [1500.0, 4000.0, 2800.0, 300.0, -0.1]
    d[14, 2:4] = [1200.0, 1400.0]
    nms_route('grid')
    got = rnms(t(d), 0.5)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 4 and np.array_equal(got, oracle.rnms(d, 0.5))
    gg = rng.integers(0, 3, len(d)).astype(np.int32)
    tt = rng.choice([0.2, 0.5], len(d)).astype(np.float32)
    got = rnms(t(d), 0.5, groups=gg, thrs=tt)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 4 and np.array_equal(got, oracle.rnms(d, 0.5, groups=gg, thrs=tt))
    # more big boxes than the big list holds (1024): the route gives up, the call is repeated on the sorted route
    d2 = oracle.synth_boxes(6000, 6000, 10)
    d2[:1100, 2] = 2000.0 + rng.random(1100).astype(np.float32) * 1500.0
    d2[:1100, 3] = 1800.0 + rng.random(1100).astype(np.float32) * 1500.0
    got = rnms(t(d2), 0.5)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 5 and np.array_equal(got, oracle.rnms(d2, 0.5))
    # the test hook: the route gives up at once, the call is repeated on the sorted route
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 2)
    got = rnms(t(d), 0.5)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 5 and np.array_equal(got, oracle.rnms(d, 0.5))
    # other kinds, per-box thresholds, strict
    d = oracle.synth_boxes(6000, 8000, 3)
    p = np.concatenate([oracle.rbox_to_poly(d).astype(np.float32), d[:, 5:6]], 1)
    h = np.stack([d[:, 0] - d[:, 2], d[:, 1] - d[:, 3], d[:, 0] + d[:, 2], d[:, 1] + d[:, 3], d[:, 5]], 1).astype(np.float32)
    h[:, :4] = np.round(h[:, :4])
    thrs = rng.choice([0.2, 0.5, 0.7], len(d)).astype(np.float32)
    nms_route('grid')
    assert np.array_equal(poly_nms(t(p), 0.4)[1].cpu().numpy(), oracle.poly_nms(p, 0.4)) and L.obb_nms_last_route() == 4
    assert np.array_equal(nms(t(h), 0.5)[1].cpu().numpy(), oracle.hbb_nms(h, 0.5)) and L.obb_nms_last_route() == 4
    assert np.array_equal(rnms(t(d), 0.5, thrs=thrs, strict=True)[1].cpu().numpy(), oracle.rnms(d, 0.5, thrs=thrs, strict=True))
    # ... and the other kinds with boxes for the big list (polygons, horizontal boxes incl. a malformed one and two huge twins)
    d2 = d.copy()
    d2[20, 2:5] = [3500.0, 2600.0, -0.4]
    d2[21, :5] = d2[20, :5] + np.float32([10.0, 5.0, 0.0, 0.0, 0.01])
    d2[22, 2:4] = [900.0, 5000.0]
    p2 = np.concatenate([oracle.rbox_to_poly(d2).astype(np.float32), d2[:, 5:6]], 1)
    h2 = h.copy()
    h2[30, :4] = [100.0, 200.0, 4100.0, 3000.0]
    h2[31, :4] = [120.0, 210.0, 4105.0, 3010.0]
    h2[32, :4] = [5000.0, 5000.0, 4000.0, 7000.0]                # x2 < x1: malformed, isolated
    assert np.array_equal(poly_nms(t(p2), 0.4)[1].cpu().numpy(), oracle.poly_nms(p2, 0.4)) and L.obb_nms_last_route() == 4
    assert np.array_equal(nms(t(h2), 0.5)[1].cpu().numpy(), oracle.hbb_nms(h2, 0.5)) and L.obb_nms_last_route() == 4
    # grouped calls take it too (test_grid_route_serves_grouped_calls)
    g = rng.integers(0, 5, len(d)).astype(np.int32)
    assert np.array_equal(rnms(t(d), 0.5, groups=g)[1].cpu().numpy(), oracle.rnms(d, 0.5, groups=g)) and L.obb_nms_last_route() == 4
    # a crowded scene: 20 000 boxes on a 600 px canvas
    dense = oracle.synth_boxes(20000, 600, 5)
    nms_route('grid')
    got = rnms(t(dense), 0.5)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 5
    assert np.array_equal(got, oracle.rnms(dense, 0.5))
    _run_nms(t(dense), _lib.KIND_RBOX, 0.5)
    assert L.obb_nms_last_route() == 2            # (the thread remembers: straight to the sorted route ...
    nms_route('auto')
    _run_nms(t(dense), _lib.KIND_RBOX, 0.5)
    assert L.obb_nms_last_route() == 5            #  ... until the knob is touched: that clears the thread's notes)
    nms_route('auto')


def test_grid_route_serves_grouped_calls(nms_route):
    """Grouped calls share the grid's table (a group's cells = the spatial grid shifted by an offset derived from a
    16-bit tag of the group id; entries of different tags are never paired, groups that share a tag are told apart by
    the records): a few groups on one canvas, thousands of groups on a small canvas each (the per-patch NMS of a tiled
    pass), negative and huge group ids, per-box thresholds, 150k boxes (auto routing, compaction in its own launch from
    2^18 boxes) -- all index-exact with the sorted route and, where it finishes in seconds, the oracle."""
    from aidet_amd import _lib, rnms
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n, canvas, ngroups in ((6000, 8000, 5), (40000, 1024, 3000), (150000, 4000, 240), (300000, 30000, 7)):
        d = oracle.synth_boxes(n, canvas, 11 + ngroups)
        g = rng.integers(0, ngroups, n).astype(np.int32)
        if ngroups == 5:
            g = (g.astype(np.int64) * 715827883 - 2 ** 31 + 17).astype(np.int32)      # ids all over int32, negative ones too
        thrs = rng.choice([0.3, 0.5], n).astype(np.float32) if ngroups != 240 else None
        nms_route('grid' if n < 150000 else 'auto')
        got = rnms(t(d), 0.5, groups=g, thrs=thrs)[1].cpu().numpy()
        route = L.obb_nms_last_route()
        nms_route('sorted')
        want = rnms(t(d), 0.5, groups=g, thrs=thrs)[1].cpu().numpy()
        assert L.obb_nms_last_route() == 2
        assert route == 4, (n, route)
        assert np.array_equal(got, want), n
        if n <= 40000:
            assert np.array_equal(got, oracle.rnms(d, 0.5, groups=g, thrs=thrs)), n
    # groups whose 16-bit tags collide (ids 65536 apart hash to the same tag only by accident: force it with equal ids
    # modulo the hash's period -- here simply two groups inside every tag by using 200 000 distinct ids)
    d = oracle.synth_boxes(30000, 3000, 2)
    g = rng.integers(0, 200000, len(d)).astype(np.int32)
    nms_route('grid')
    got = rnms(t(d), 0.4, groups=g)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 4 and np.array_equal(got, oracle.rnms(d, 0.4, groups=g))
    # a crowded grouped scene gives the route up: repeated on the sorted route, and remembered for GROUPED calls of the
    # size class only
    dense = oracle.synth_boxes(20000, 500, 8)
    g = rng.integers(0, 2, len(dense)).astype(np.int32)
    got = rnms(t(dense), 0.5, groups=g)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 5 and np.array_equal(got, oracle.rnms(dense, 0.5, groups=g))
    nms_route('auto')


def test_fused_tiled_phases_on_the_grid_route_and_its_repeat(nms_route):
    """A tiled pass whose two NMS calls are large enough for the grid route (forced here: the scene is a small one) ==
    the same pass with the grid off == the step-by-step pass; a pass whose per-patch NMS gives the grid up (every
    detection of a patch in one spot: a crowded cell) is repeated by the driver (K word OBB_TILED_REPEAT / OBB_ERETRY)
    and ends with the sorted route's result."""
    from aidet_amd import _lib, dota_merge
    L = _lib.lib()
    args, scene = _tiled_args(images=2, image_size=4096, objects=6000, classes=5, seed=7)
    thrs = [0.3, 0.5, 0.1, 0.4, 0.2]
    nms_route('grid')
    gk, gkeep, grows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs)
    nms_route('sorted')
    sk, skeep, srows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs)
    try:
        dota_merge.FUSED = False
        pk, pkeep, prows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs)
    finally:
        dota_merge.FUSED = True
    assert torch.equal(grows, srows) and torch.equal(gkeep, skeep) and torch.equal(gk, sk)
    assert torch.equal(grows, prows) and torch.equal(gkeep, pkeep) and torch.equal(gk, pk)
    assert 0 < gk.size(0) < grows.size(0) < args[0].size(0)
    # crowded: 400 detections of one class within a few pixels in every patch -- more than a cell holds
    rng = np.random.default_rng(3)
    n, P = 20000, 50
    d = np.zeros((n, 6), np.float32)
    d[:, 0:2] = 300.0 + rng.random((n, 2)).astype(np.float32) * 8.0
    d[:, 2:4] = 20.0 + rng.random((n, 2)).astype(np.float32) * 4.0
    d[:, 4] = -rng.random(n).astype(np.float32)
    d[:, 5] = (rng.permutation(n) + 0.5) / n
    patch = np.repeat(np.arange(P, dtype=np.int64), n // P)
    base = [t(d), t(np.zeros(n, np.int64)), t(patch), t(np.arange(n, dtype=np.int64)),
            t((rng.integers(0, 8, (P, 2)) * 512).astype(np.float32)), t(np.zeros(P, np.int64))]
    dota_merge._phase_capacity.clear()
    nms_route('sorted')
    want = dota_merge.tiled_inference_merge(*base, 3, 0.5, 0.3, None)
    nms_route('grid')
    got = dota_merge.tiled_inference_merge(*base, 3, 0.5, 0.3, None)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # the repeat paths of both phases through the hook (OBB_TUNE_NMS_GRID = 2: the route gives up at once)
    before = dota_merge.phase_repeats
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 2)
    hk, hkeep, hrows = dota_merge.tiled_inference_merge(*args, 5, 0.5, 0.3, thrs)
    assert dota_merge.phase_repeats == before + 2               # the per-patch phase and the merge phase, once each
    assert torch.equal(hrows, srows) and torch.equal(hkeep, skeep) and torch.equal(hk, sk)
    nms_route('auto')
    dota_merge._phase_capacity.clear()


def test_grid_route_boxes_clamped_into_the_frame(nms_route):
    """The quantisation frame comes from a sample (and trims far outliers): boxes outside it are CLAMPED into its first /
    last cells, where their cells say nothing about their extent -- the bound on the quantised overlap must not be applied
    to them (a chain of 2000 boxes along x of which the frame kernel once sampled the first 1024 only; a cluster of
    mutually overlapping boxes far outside the scene).  Crowded edge cells also exercise the overflow list."""
    from aidet_amd import _lib, rnms
    L = _lib.lib()
    for n, step in ((2000, 4.0), (4000, 4.0), (2000, 3.0), (3000, 2.5)):
        d = np.zeros((n, 6), np.float32)
        d[:, 0] = np.arange(n) * step; d[:, 1] = 5; d[:, 2] = 10; d[:, 3] = 10
        d[:, 5] = 1.0 - np.arange(n) / n
        nms_route('grid')
        got = rnms(t(d), 0.4)[1].cpu().numpy()
        assert L.obb_nms_last_route() in (4, 5)
        assert np.array_equal(got, oracle.rnms(d, 0.4)), (n, step)
    rng = np.random.default_rng(12)
    d = oracle.synth_boxes(5000, 6000, 21)
    far = np.zeros((60, 6), np.float32)
    far[:, 0:2] = 3.0e5 + rng.random((60, 2)).astype(np.float32) * 30.0
    far[:, 2:4] = 40.0 + rng.random((60, 2)).astype(np.float32) * 20.0
    far[:, 4] = -rng.random(60).astype(np.float32)
    far[:, 5] = rng.random(60).astype(np.float32)
    d = np.concatenate([d, far])[rng.permutation(5060)]
    nms_route('grid')
    got = rnms(t(d), 0.3)[1].cpu().numpy()
    assert L.obb_nms_last_route() == 4
    assert np.array_equal(got, oracle.rnms(d, 0.3))
    nms_route('auto')


def test_position_free_soft_nms_and_its_fallback():
    """`soft_nms` on one float array up to 4096 boxes: the kernel without positions serves every call whose maxima are
    unique (index-exact with the oracle's array loop, scores bit-exact for the linear method), gives up on a shared
    maximum / a NaN / a score that decays into a NaN, and the array kernel then replays the permutation -- same result as
    with the first try switched off (knob 2).  All launch shapes (1 .. 16 boxes per thread), heavy drops, -0 scores."""
    from aidet_amd import _lib, soft_nms
    L = _lib.lib()
    rng = np.random.default_rng(23)
    for n, canvas, min_score in ((1, 50, 0.01), (2, 10, 0.3), (200, 100, 0.3), (257, 120, 0.05), (700, 200, 0.2), (1024, 200, 0.05),
                                 (2000, 300, 0.05), (2048, 260, 0.3), (3000, 400, 0.1), (4096, 500, 0.05)):
        x1, y1 = rng.uniform(0, canvas, n), rng.uniform(0, canvas, n)
        d = np.stack([x1, y1, x1 + rng.uniform(4, 60, n), y1 + rng.uniform(4, 60, n), rng.permutation(n) / n + 1e-3], 1).astype(np.float32)
        if n > 10:
            d[7, 4] = -0.0
        for method, thr in (('linear', 0.3), ('gaussian', 0.3), ('linear', 0.0)):
            nd, inds = soft_nms(t(d), thr, method=method, sigma=0.5, min_score=min_score)
            path = L.obb_soft_nms_last_path()
            rd, ri = oracle.soft_nms(d, thr, method, 0.5, min_score)
            assert np.array_equal(inds.cpu().numpy(), ri), (n, method, thr)
            assert np.array_equal(nd.cpu().numpy(), rd), (n, method)      # (gaussian too: glibc's expf restated on the device)
            assert path in (2, 3)
            if n >= 200 and method == 'linear' and thr == 0.3:
                assert path == 2, (n, path)       # distinct scores, distinct decays: no shared maximum
        # shared maxima, NaN scores, a NaN corner (its overlaps, hence the decayed scores of its neighbours' gaussian weights, are NaN)
        e = d.copy()
        e[::5, 4] = np.float32(0.5)
        if n > 10:
            e[3, 4] = np.nan
        f = d.copy()
        if n > 10:
            f[5, 0] = np.nan
        for arr in (e, f):
            for method in ('linear', 'gaussian'):
                nd, inds = soft_nms(t(arr), 0.3, method=method, sigma=0.5, min_score=min_score)
                path = L.obb_soft_nms_last_path()
                rd, ri = oracle.soft_nms(arr, 0.3, method, 0.5, min_score)
                assert np.array_equal(inds.cpu().numpy(), ri), (n, method, path)
                assert np.array_equal(nd.cpu().numpy(), rd, equal_nan=True), (n, method)
                try:
                    _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, 2)
                    nd2, inds2 = soft_nms(t(arr), 0.3, method=method, sigma=0.5, min_score=min_score)
                    assert L.obb_soft_nms_last_path() == 0
                finally:
                    _lib.set_tuning(_lib.TUNE_SOFT_NMS_REG, -1)
                assert torch.equal(inds, inds2) and torch.equal(nd.nan_to_num(-7.0), nd2.nan_to_num(-7.0))
        if n >= 200:
            nd, inds = soft_nms(t(e), 0.3, method='linear', sigma=0.5, min_score=min_score)
            assert L.obb_soft_nms_last_path() == 3
    # the other launch shapes of the kernel (OBB_TUNE_SOFT_NMS_SHAPE: 256 / 512 threads x more boxes; measurements only)
    for shape in (1, 2):
        try:
            _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, shape)
            for n in (300, 900, 2048, 3500):
                x1, y1 = rng.uniform(0, 400, n), rng.uniform(0, 400, n)
                d = np.stack([x1, y1, x1 + rng.uniform(4, 60, n), y1 + rng.uniform(4, 60, n), rng.permutation(n) / n + 1e-3], 1).astype(np.float32)
                nd, inds = soft_nms(t(d), 0.3, method='linear', sigma=0.5, min_score=0.05)
                rd, ri = oracle.soft_nms(d, 0.3, 'linear', 0.5, 0.05)
                assert np.array_equal(inds.cpu().numpy(), ri) and np.array_equal(nd.cpu().numpy(), rd), (shape, n)
        finally:
            _lib.set_tuning(_lib.TUNE_SOFT_NMS_SHAPE, -1)


def test_grid_route_on_millions_of_boxes(nms_route):
    """3 M boxes -- ungrouped on a canvas of the headline's density, and in 5000 groups overlaid on a smaller one: the
    grid route (a table of 2^21 / 2^22 cells, compaction in its own launch) returns the sorted route's keep list."""
    import workloads
    from aidet_amd import _lib
    from aidet_amd.nms_wrapper import _run_nms
    L = _lib.lib()
    n = 3000000
    rng = np.random.default_rng(4)
    for ng, canvas in ((0, 110000.0), (5000, 12000.0)):
        d = t(workloads.boxes(n, canvas, 5))
        g = None if ng == 0 else t(rng.integers(0, ng, n).astype(np.int32))
        nms_route('auto')
        keep, _ = _run_nms(d, _lib.KIND_RBOX, 0.5, g)
        assert L.obb_nms_last_route() == 4
        nms_route('listed')
        want, _ = _run_nms(d, _lib.KIND_RBOX, 0.5, g)
        assert L.obb_nms_last_route() == 2
        assert torch.equal(keep, want) and 0 < keep.numel() < n
        del d, g, keep, want
        torch.cuda.empty_cache()
    nms_route('auto')
