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
