"""Round 6: the NMS on a RETAINED workspace (OBB_NMS_RETAIN, include/obbops.h) -- three launches instead of four on the
grid route: the previous call's tail has cleared the cell counters and left the frame of its own boxes.  Whatever the
state of the workspace, the keep list must be the oracle's, index for index (nms_cpu.cpp:21-59 semantics).

Every call goes through the C ABI with a workspace this file owns, so that the tests can break the caller's promise on
purpose (another writer, another thread, another size class) and watch the library notice."""
import ctypes
import threading

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


class Caller(object):
    """obb_nms_ex_f32 on a workspace of its own"""

    def __init__(self, nmax, kind=None, capacity=0):
        from aidet_amd import _lib
        self._lib = _lib
        self.L = _lib.lib()
        self.kind = _lib.KIND_RBOX if kind is None else kind
        # (an explicit pair capacity: the default follows the calling thread's capacity hints, which other tests leave behind)
        nbytes = self.L.obb_nms_workspace_bytes(nmax, self.kind, capacity or max(1 << 18, 64 * nmax))
        assert nbytes > 0
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        self.ws.random_(0, 255)            # a fresh workspace holds anything
        self.info = (ctypes.c_int64 * 4)()

    def __call__(self, d, thr, retain=True, groups=None, thrs=None, strict=False):
        _lib, L = self._lib, self.L
        dt = d if torch.is_tensor(d) else torch.from_numpy(np.ascontiguousarray(d)).to(dev())
        n = dt.size(0)
        keep = torch.empty(n, dtype=torch.int64, device=dev())
        gp = tp = None
        if groups is not None:
            groups = torch.as_tensor(groups, dtype=torch.int32).to(dev())
            gp = groups.data_ptr()
        if thrs is not None:
            thrs = torch.as_tensor(thrs, dtype=torch.float32).to(dev())
            tp = thrs.data_ptr()
        flags = (_lib.NMS_RETAIN if retain else 0) | (_lib.NMS_STRICT if strict else 0)
        assert L.obb_nms_workspace_bytes(n, self.kind, max(1 << 18, 64 * n)) <= self.ws.numel()
        rc = L.obb_nms_ex_f32(dt.data_ptr(), n, self.kind, float(thr), gp, tp, flags, keep.data_ptr(), self.info,
                              self.ws.data_ptr(), self.ws.numel(), _lib.current_stream_handle(dev()))
        _lib.check(rc)
        return keep[:self.info[0]].cpu().numpy(), int(L.obb_nms_last_retained()), int(L.obb_nms_last_route())


@pytest.fixture
def grid_everywhere():
    """the grid route wherever it applies (n >= 256), as the parity tests force it"""
    from aidet_amd import _lib
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
    yield
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, -1)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, -1)
    _lib.set_tuning(_lib.TUNE_NMS_RETAIN, -1)


def test_a_stream_of_scenes_runs_on_three_launches():
    """the bench's situation: scenes of one kind, one after the other -- the first call seeds, the others run on what
    their predecessor left; every keep list is the oracle's"""
    call = Caller(30000)
    modes = []
    for seed in range(6):
        d = oracle.synth_boxes(30000, 8000, 40 + seed)
        got, mode, route = call(d, 0.5)
        assert route == 4
        assert np.array_equal(got, oracle.rnms(d, 0.5)), 'scene %d (retained %d)' % (seed, mode)
        modes.append(mode)
    assert modes == [1, 2, 2, 2, 2, 2]
    # the same scenes without the flag: four launches, the same lists
    plain = Caller(30000)
    d = oracle.synth_boxes(30000, 8000, 41)
    a, mode, _ = plain(d, 0.5, retain=False)
    b, _, _ = call(d, 0.5)
    assert mode == 0 and np.array_equal(a, b)


def test_the_flag_can_be_switched_off():
    from aidet_amd import _lib
    call = Caller(20000)
    d = oracle.synth_boxes(20000, 6000, 5)
    ref = oracle.rnms(d, 0.5)
    _lib.set_tuning(_lib.TUNE_NMS_RETAIN, 0)
    try:
        for _ in range(2):
            got, mode, _r = call(d, 0.5)
            assert mode == 0 and np.array_equal(got, ref)
    finally:
        _lib.set_tuning(_lib.TUNE_NMS_RETAIN, -1)
    assert [call(d, 0.5)[1] for _ in range(2)] == [1, 2]


def test_a_scene_unlike_its_predecessor(grid_everywhere):
    """the frame a call finds is its predecessor's: a scene somewhere else, of another scale or another density must
    come out the same -- on a crowded grid by the repeat with a frame of its own, at worst on the sorted route"""
    n = 12000
    call = Caller(n)
    scenes = []
    a = oracle.synth_boxes(n, 7000, 1)
    scenes.append(('compact', a))
    b = oracle.synth_boxes(n, 60000, 2)
    b[:, :2] += 500000.0                               # far away from the first frame: everything clamps into one corner
    scenes.append(('far away, sparse', b))
    c = oracle.synth_boxes(n, 7000, 3)
    c[:, 2:4] *= 0.1                                   # tiny boxes: the cells are far too wide
    scenes.append(('tiny boxes', c))
    e = oracle.synth_boxes(n, 30000, 4)
    e[:, 2:4] *= 12.0                                  # large boxes: the predecessor's cells are far too narrow (big list / many cells)
    scenes.append(('large boxes', e))
    scenes.append(('compact again', a))
    f = a.copy()
    f[:, 0] = f[:, 0] * 1e-3 + 77.0                    # a vertical line: one axis of the old frame is almost empty
    scenes.append(('a line', f))
    g = oracle.synth_boxes(n, 700, 9)                  # dense: crowded cells with any frame -> the sorted route
    scenes.append(('dense', g))
    scenes.append(('compact after dense', a))
    scenes.append(('compact once more', oracle.synth_boxes(n, 7000, 8)))
    seen = []
    from aidet_amd import _lib
    for name, d in scenes:
        _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)         # (forgets "this size class gave the grid up": every scene tries it)
        got, mode, route = call(d, 0.5)
        ref = oracle.rnms(d, 0.5)
        assert np.array_equal(got, ref), '%s (retained %d, route %d)' % (name, mode, route)
        seen.append((name, mode, route))
    # (a scene unlike its predecessor is repeated with a frame of its own: retained 1; one like it runs on three launches)
    assert seen[0][1] == 1 and seen[-1][1:] == (2, 4), seen


def test_somebody_else_wrote_the_workspace():
    """the header is checked by every launch of a three-launch call: a workspace whose header changed makes the call leave
    without touching anything and run again seeded"""
    n = 15000
    call = Caller(n)
    d = [oracle.synth_boxes(n, 5000, 20 + k) for k in range(4)]
    ref = [oracle.rnms(x, 0.4) for x in d]
    assert call(d[0], 0.4)[1] == 1
    assert call(d[1], 0.4)[1] == 2
    call.ws[:256].zero_()                              # the header is gone (another tensor lived here)
    got, mode, _r = call(d[2], 0.4)
    assert mode == 1 and np.array_equal(got, ref[2])
    got, mode, _r = call(d[3], 0.4)
    assert mode == 2 and np.array_equal(got, ref[3])
    call.ws.random_(0, 255)                            # everything is gone
    got, mode, _r = call(d[0], 0.4)
    assert mode == 1 and np.array_equal(got, ref[0])
    got, mode, _r = call(d[1], 0.4)
    assert mode == 2 and np.array_equal(got, ref[1])


def test_other_calls_on_the_same_workspace(grid_everywhere):
    """calls of other size classes, without the flag, on other routes and through the asynchronous entry point end what the
    thread believes about the workspace: the next retained call seeds it again"""
    from aidet_amd import _lib
    L = _lib.lib()
    call = Caller(40000)
    big = oracle.synth_boxes(40000, 9000, 1)
    small = oracle.synth_boxes(3000, 2500, 2)           # another table geometry: its arrays lie where the big call's counters are
    mid = oracle.synth_boxes(33000, 9000, 3)            # the same geometry as `big`
    ref = {k: oracle.rnms(v, 0.5) for k, v in (('big', big), ('small', small), ('mid', mid))}
    assert call(big, 0.5)[1] == 1
    got, mode, _r = call(mid, 0.5)
    assert mode == 2 and np.array_equal(got, ref['mid'])            # n changed, the class did not
    got, mode, _r = call(small, 0.5)
    assert mode == 1 and np.array_equal(got, ref['small'])
    got, mode, _r = call(big, 0.5)
    assert mode == 1 and np.array_equal(got, ref['big'])            # the small call's arrays overwrote the counters
    got, mode, _r = call(big, 0.5)
    assert mode == 2 and np.array_equal(got, ref['big'])
    got, mode, _r = call(mid, 0.5, retain=False)                      # a plain call: four launches, the note is dropped
    assert mode == 0 and np.array_equal(got, ref['mid'])
    got, mode, _r = call(big, 0.5)
    assert mode == 1 and np.array_equal(got, ref['big'])
    # the sorted route in between
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 0)
    got, mode, route = call(mid, 0.5)
    assert mode == 0 and route == 2 and np.array_equal(got, ref['mid'])
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
    got, mode, route = call(big, 0.5)
    assert mode == 1 and route == 4 and np.array_equal(got, ref['big'])
    # the asynchronous entry point refuses the flag and forgets the workspace
    dt = torch.from_numpy(mid).to(dev())
    keep = torch.empty(len(mid), dtype=torch.int64, device=dev())
    rec = torch.zeros(8, dtype=torch.int64, device=dev())
    st = _lib.current_stream_handle(dev())
    assert L.obb_nms_async_f32(dt.data_ptr(), len(mid), _lib.KIND_RBOX, 0.5, None, None, _lib.NMS_RETAIN, keep.data_ptr(),
                               rec.data_ptr(), call.ws.data_ptr(), call.ws.numel(), st) == _lib.OBB_EINVAL
    assert L.obb_nms_async_f32(dt.data_ptr(), len(mid), _lib.KIND_RBOX, 0.5, None, None, 0, keep.data_ptr(),
                               rec.data_ptr(), call.ws.data_ptr(), call.ws.numel(), st) == _lib.OBB_OK
    torch.cuda.synchronize()
    assert np.array_equal(keep[:int(rec[0])].cpu().numpy(), ref['mid'])
    got, mode, _r = call(big, 0.5)
    assert mode == 1 and np.array_equal(got, ref['big'])


def test_two_threads_one_workspace():
    """breaking the promise: two threads take turns on ONE workspace.  Each thread's note of the header goes stale with the
    other's call; the device-side check turns that into a repeated call, never into a wrong list"""
    n = 14000
    call = Caller(n)
    d = [oracle.synth_boxes(n, 5000, 60 + k) for k in range(6)]
    ref = [oracle.rnms(x, 0.5) for x in d]
    out = {}

    def other(k):
        torch.cuda.set_device(0)
        out[k] = call(d[k], 0.5)

    for k in range(6):
        if k % 2:
            th = threading.Thread(target=other, args=(k,))
            th.start()
            th.join()
        else:
            out[k] = call(d[k], 0.5)
    for k in range(6):
        assert np.array_equal(out[k][0], ref[k]), (k, out[k][1:])
    # each side believed the state its own last call left: the header had moved on, the three-launch call left, the call was seeded
    assert [out[k][1] for k in range(6)] == [1, 1, 1, 1, 1, 1]


def test_groups_thresholds_kinds_and_the_strict_rule(grid_everywhere):
    from aidet_amd import _lib
    n = 16000
    rng = np.random.default_rng(5)
    call = Caller(n)
    for rep in range(3):
        d = oracle.synth_boxes(n, 8000, 70 + rep)
        groups = rng.integers(0, 7, n).astype(np.int32)
        table = np.array([0.3, 0.5, 0.1, 0.4, 0.2, 0.6, 0.45], np.float32)
        thrs = table[groups]
        got, mode, route = call(d, 0.5, groups=groups, thrs=thrs)
        assert np.array_equal(got, oracle.rnms(d, 0.5, groups=groups, thrs=thrs)), (rep, mode, route)
        assert mode == (1 if rep == 0 else 2)
    got, mode, _r = call(d, 0.5)                            # ungrouped on the grouped calls' state: another signature
    assert mode == 1 and np.array_equal(got, oracle.rnms(d, 0.5))
    got, mode, _r = call(d, 0.5, strict=True)
    assert mode == 2 and np.array_equal(got, oracle.rnms(d, 0.5, strict=True))
    # polygons and horizontal boxes
    poly = Caller(n, kind=_lib.KIND_POLY)
    for rep in range(2):
        dd = oracle.synth_boxes(n, 8000, 80 + rep)
        p = np.concatenate([oracle.rbox_to_poly(dd[:, :5]), dd[:, 5:6]], 1).astype(np.float32)
        got, mode, _r = poly(p, 0.5)
        assert mode == rep + 1 and np.array_equal(got, oracle.poly_nms(p, 0.5))
    hbb = Caller(n, kind=_lib.KIND_HBB)
    for rep in range(2):
        dd = oracle.synth_boxes(n, 8000, 90 + rep)
        h = np.stack([dd[:, 0] - dd[:, 2] / 2, dd[:, 1] - dd[:, 3] / 2, dd[:, 0] + dd[:, 2] / 2, dd[:, 1] + dd[:, 3] / 2,
                      dd[:, 5]], 1).astype(np.float32)
        got, mode, _r = hbb(h, 0.5)
        assert mode == rep + 1 and np.array_equal(got, oracle.hbb_nms(h, 0.5))


def test_overflowing_pair_lists_on_a_retained_workspace():
    """a call whose pair lists overflow returns OBB_EWORKSPACE with its tail run: the workspace stays usable"""
    from aidet_amd import _lib
    L = _lib.lib()
    n = 12000
    d = oracle.synth_boxes(n, 600, 3)                      # dense: far more pairs than a small capacity holds
    dt = torch.from_numpy(d).to(dev())
    nbytes = L.obb_nms_workspace_bytes(n, _lib.KIND_RBOX, 1 << 14)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    keep = torch.empty(n, dtype=torch.int64, device=dev())
    info = (ctypes.c_int64 * 4)()
    st = _lib.current_stream_handle(dev())
    _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, 0)
    _lib.set_tuning(_lib.TUNE_NMS_GRID, 1)
    try:
        rcs = [L.obb_nms_ex_f32(dt.data_ptr(), n, _lib.KIND_RBOX, 0.5, None, None, _lib.NMS_RETAIN, keep.data_ptr(), info,
                                ws.data_ptr(), nbytes, st) for _ in range(3)]
        assert all(rc in (_lib.OBB_EWORKSPACE, _lib.OBB_OK) for rc in rcs) and _lib.OBB_EWORKSPACE in rcs
        sparse = oracle.synth_boxes(n, 9000, 4)
        st2 = torch.from_numpy(sparse).to(dev())
        for _ in range(2):
            rc = L.obb_nms_ex_f32(st2.data_ptr(), n, _lib.KIND_RBOX, 0.5, None, None, _lib.NMS_RETAIN, keep.data_ptr(), info,
                                  ws.data_ptr(), nbytes, st)
            assert rc == _lib.OBB_OK
            assert np.array_equal(keep[:info[0]].cpu().numpy(), oracle.rnms(sparse, 0.5))
    finally:
        _lib.set_tuning(_lib.TUNE_NMS_DIRECT_MAX, -1)
        _lib.set_tuning(_lib.TUNE_NMS_GRID, -1)


def test_the_wrapper_keeps_its_workspace_and_runs_on_three_launches():
    """aidet_amd.rnms (the binding owns the workspace it passes): the second call of a size class is a three-launch one"""
    from aidet_amd import _lib, rnms
    L = _lib.lib()
    seen = []
    for seed in range(4):
        d = oracle.synth_boxes(25000, 7000, seed)
        _, inds = rnms(torch.from_numpy(d).to(dev()), 0.5)
        seen.append(int(L.obb_nms_last_retained()))
        assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.5))
    assert seen[1:] == [2, 2, 2], seen


def test_headline_scene_on_a_retained_workspace():
    """BASELINE.json configs[1] on the three-launch call (the bench's timed calls): bit-exact against the oracle"""
    from tests.test_gpu_parity import _cfg2_reference
    call = Caller(100000)
    d, ref = _cfg2_reference(20000, 3)
    warm = oracle.synth_boxes(100000, 20000, 11)
    assert call(warm, 0.5)[1] == 1
    got, mode, route = call(d, 0.5)
    assert mode == 2 and route == 4 and np.array_equal(got, ref)


def test_release_workspaces_starts_over():
    from aidet_amd import _lib, nms_wrapper, rnms
    L = _lib.lib()
    d = oracle.synth_boxes(21000, 6500, 17)
    ref = oracle.rnms(d, 0.5)
    dt = torch.from_numpy(d).to(dev())
    for _ in range(2):
        assert np.array_equal(rnms(dt, 0.5)[1].cpu().numpy(), ref)
    assert int(L.obb_nms_last_retained()) == 2
    nms_wrapper.release_workspaces()
    assert np.array_equal(rnms(dt, 0.5)[1].cpu().numpy(), ref)
    assert int(L.obb_nms_last_retained()) == 1
    assert np.array_equal(rnms(dt, 0.5)[1].cpu().numpy(), ref)
    assert int(L.obb_nms_last_retained()) == 2


def test_four_threads_four_streams_at_once():
    """the library keeps its state per thread and per workspace: four threads, each with a stream and a workspace of its own,
    run retained calls AT THE SAME TIME (ctypes releases the GIL for the call) -- every list is the oracle's"""
    n = 18000
    scenes = [oracle.synth_boxes(n, 6500, 300 + k) for k in range(8)]
    refs = [oracle.rnms(d, 0.5) for d in scenes]
    errors, modes = [], {}
    start = threading.Barrier(4)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream(device=dev())
            with torch.cuda.stream(st):
                call = Caller(n)
                ds = [torch.from_numpy(d).to(dev()) for d in scenes]
                torch.cuda.synchronize()
                start.wait()
                seen = []
                for rep in range(40):
                    k = (rep * 3 + t) % 8
                    got, mode, _route = call(ds[k], 0.5)
                    seen.append(mode)
                    if not np.array_equal(got, refs[k]):
                        errors.append((t, rep, k, mode))
                modes[t] = seen
        except Exception as e:      # (an assertion in a thread would vanish)
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    for t in range(4):
        assert modes[t][0] == 1 and all(m == 2 for m in modes[t][1:]), (t, modes[t])
