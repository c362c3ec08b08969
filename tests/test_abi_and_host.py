"""CPU tests: the C-ABI library loads and exports every symbol include/obbops.h
declares; argument validation; host-side wrapper behaviour that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'obbops.h')).read()
    return re.findall(r'OBB_API\s+[\w\s\*]+?\b(obb_\w+)\s*\(', text)


def test_library_exports_every_declared_symbol():
    from aidet_amd import _lib
    L = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), 'libobbops.so does not export ' + n
    assert L.obb_version() >= 100
    assert b'gfx950' in L.obb_build_info()


def test_header_cites_reference_for_each_entry_point():
    text = open(os.path.join(ROOT, 'include', 'obbops.h')).read()
    for cite in ('geometry.py:4-88', 'nms_cuda.cpp:8-17', 'nms_cpu.cpp', 'roi_align_cuda.cpp',
                 'transforms.py:45-55'):
        assert cite in text


def test_argument_validation_without_gpu():
    """Validation happens before any device work, so it is testable here."""
    from aidet_amd import _lib
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    assert L.obb_overlaps_f32(null, -1, null, 3, 0, 0, 0, null, null) == _lib.OBB_EINVAL
    assert b'negative' in L.obb_last_error()
    assert L.obb_overlaps_f32(null, 2, null, 3, 7, 0, 0, null, null) == _lib.OBB_EINVAL
    assert L.obb_overlaps_f32(null, 2, null, 3, 0, 9, 0, null, null) == _lib.OBB_EINVAL
    assert L.obb_overlaps_f32(null, 2, null, 3, 0, 0, 1, null, null) == _lib.OBB_EINVAL   # aligned, m != n
    assert L.obb_overlaps_f32(null, 0, null, 3, 0, 0, 0, null, null) == _lib.OBB_OK       # empty: no-op
    assert L.obb_overlaps_f32(null, 2, null, 3, 0, 0, 0, null, null) == _lib.OBB_EINVAL   # null pointers
    info = (ctypes.c_int64 * 4)()
    assert L.obb_nms_f32(null, -5, 0, 0.5, null, info, null, 0, null) == _lib.OBB_EINVAL
    assert L.obb_nms_f32(null, 0, 0, 0.5, null, info, null, 0, null) == _lib.OBB_OK
    assert L.obb_nms_f32(null, 5, 3, 0.5, null, info, null, 0, null) == _lib.OBB_EINVAL
    assert L.obb_rroi_align_forward_f32(null, 1, 1, 4, 4, null, 1, 0, 7, 1.0, 2, 1.0, null, null) == _lib.OBB_EINVAL
    assert L.obb_rroi_align_forward_f32(null, 1, 1, 4, 4, null, 0, 7, 7, 1.0, 2, 1.0, null, null) == _lib.OBB_OK
    assert L.obb_rbox_to_poly_f32(null, 0, null, null) == _lib.OBB_OK
    buf = (ctypes.c_float * 7)()
    assert L.obb_nms_profile_last(buf) == _lib.OBB_EINVAL   # nothing profiled yet
    # round 2 entry points
    one = ctypes.c_void_p(256)   # (never dereferenced: validation fails first)
    assert L.obb_nms_ex_f32(one, 5, 0, 0.5, null, null, 8, one, info, one, 1 << 20, null) == _lib.OBB_EINVAL
    assert b'flags' in L.obb_last_error()
    # (horizontal boxes, `>=`, thr <= 0: refused until round 4; since round 5 served as nms_cpu.cpp:56 serves it --
    #  validation passes, tests/test_round5_gpu.py::test_hbb_nms_nonpositive_threshold_matches_reference)
    assert L.obb_nms_ex_f32(null, 0, _lib.KIND_HBB, 0.5, null, null, 0, null, info, null, 0, null) == _lib.OBB_OK
    assert L.obb_nms_async_f32(one, 5, 0, 0.5, null, null, 0, one, null, one, 1 << 20, null) == _lib.OBB_EINVAL
    assert L.obb_nms_async_result(null, 5, 1 << 20, info) == _lib.OBB_EINVAL
    # round 6: OBB_NMS_RETAIN is a flag of the synchronous obb_nms_ex_f32 only (the host reads the call's record)
    assert L.obb_nms_async_f32(one, 5, 0, 0.5, null, null, _lib.NMS_RETAIN, one, one, one, 1 << 20, null) == _lib.OBB_EINVAL
    assert b'OBB_NMS_RETAIN' in L.obb_last_error()
    assert L.obb_nms_hbb_f64(one, 5, 0.5, null, null, _lib.NMS_RETAIN, one, info, one, 1 << 20, null) == _lib.OBB_EINVAL
    assert b'OBB_NMS_RETAIN' in L.obb_last_error()
    assert L.obb_nms_ex_f32(one, 5, 0, 0.5, null, null, _lib.NMS_RETAIN | 4, one, info, one, 1 << 20, null) == _lib.OBB_EINVAL
    assert L.obb_nms_last_retained() in (0, 1, 2)
    rec = (ctypes.c_int64 * 8)()                           # all-zero record = empty result
    assert L.obb_nms_async_result(rec, 0, 0, info) == _lib.OBB_OK and info[0] == 0
    ws = L.obb_nms_workspace_bytes(1000, 0, 0)
    rec[0] = 7                                             # K = 7, fused route, no overflow
    assert L.obb_nms_async_result(rec, 1000, ws, info) == _lib.OBB_OK and info[0] == 7
    rec[1] = 1 << 40                                       # more edges than the workspace holds
    assert L.obb_nms_async_result(rec, 1000, ws, info) == _lib.OBB_EWORKSPACE and info[3] > (1 << 39)
    # a record whose route word says "the grid route gave this scene up" (3 | skip key << 8, at byte 48): the caller repeats
    # the call as it is -- OBB_ERETRY from obb_nms_async_result and, for a tiled phase's record, from obb_tiled_result
    rec2 = (ctypes.c_uint32 * 16)()
    rec2[12] = 3 | ((1 + _lib.KIND_RBOX + 4 + 8) << 8)
    assert L.obb_nms_async_result(rec2, 20000, L.obb_nms_workspace_bytes(20000, 0, 0), info) == _lib.OBB_ERETRY
    assert b'repeat' in L.obb_last_error()
    nb = L.obb_tiled_local_workspace_bytes(20000, 1, 0)
    assert L.obb_tiled_result(rec2, 0, 20000, 6, 1, -1, nb, info) == _lib.OBB_ERETRY
    assert L.obb_set_tuning(_lib.TUNE_NMS_GRID, -1) == _lib.OBB_OK          # (forgets the note the two calls above left)
    # a record whose `rounds` word (byte 16) is all ones: the resolve kernel stopped at its bound (its workgroups were not resident
    # together) -- the keep list is not valid and the entry points say so instead of returning it
    rec3 = (ctypes.c_int64 * 8)()
    rec3[0] = 7; rec3[2] = -1
    assert L.obb_nms_async_result(rec3, 1000, ws, info) == _lib.OBB_ESTALLED and info[0] == 0
    assert b'converg' in L.obb_last_error()
    cnt = ctypes.c_int64(5)
    assert L.obb_soft_nms_f32(null, -1, 0.5, 1, 0.5, 0.001, null, null, ctypes.byref(cnt), null, 0, null) == _lib.OBB_EINVAL
    assert L.obb_soft_nms_f32(null, 3, 0.5, 7, 0.5, 0.001, null, null, ctypes.byref(cnt), null, 0, null) == _lib.OBB_EINVAL
    assert L.obb_soft_nms_f32(null, 0, 0.5, 1, 0.5, 0.001, null, null, ctypes.byref(cnt), null, 0, null) == _lib.OBB_OK
    assert cnt.value == 0
    assert L.obb_soft_nms_workspace_bytes(100) == 256 and L.obb_soft_nms_workspace_bytes(5000) >= 5000 * 32
    assert L.obb_set_tuning(99, 1) == _lib.OBB_EINVAL
    assert L.obb_set_tuning(_lib.TUNE_NMS_FUSED_MAX, 64) == _lib.OBB_OK
    assert L.obb_set_tuning(_lib.TUNE_NMS_FUSED_MAX, -1) == _lib.OBB_OK


def test_library_reads_no_environment_variable():
    """Test knobs are explicit API calls (obb_set_tuning), not getenv in the entry points."""
    for f in os.listdir(os.path.join(ROOT, 'aidet_amd', 'csrc')):
        if f.endswith(('.hip', '.cuh', '.h', '.cpp')):
            assert 'getenv' not in open(os.path.join(ROOT, 'aidet_amd', 'csrc', f)).read(), f


def test_wrapper_type_errors_match_reference():
    """nms_wrapper.py:44-47 raises TypeError for anything but Tensor / ndarray."""
    from aidet_amd import nms, poly_nms, rnms
    for fn in (rnms, poly_nms, nms):
        with pytest.raises(TypeError):
            fn([[0, 0, 1, 1, 0, 1]], 0.5)
    with pytest.raises(ValueError):
        rnms(np.zeros((3, 5), np.float32), 0.5)


def test_empty_inputs_need_no_device():
    """N == 0 returns before any device work (nms_wrapper.py:50-51)."""
    from aidet_amd import rbbox_overlaps, rnms
    kept, inds = rnms(np.zeros((0, 6), np.float32), 0.5)
    assert kept.shape == (0, 6) and inds.dtype == np.int64 and inds.shape == (0,)
    kept, inds = rnms(torch.zeros((0, 6)), 0.5)
    assert inds.dtype == torch.int64 and inds.numel() == 0
    assert tuple(rbbox_overlaps(torch.zeros(0, 5), torch.zeros(3, 5)).shape) == (0, 3)
    assert tuple(rbbox_overlaps(torch.zeros(0, 5), torch.zeros(0, 5), is_aligned=True).shape) == (0, 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_fails_loudly_without_gpu():
    """No CPU fallback exists: compute calls raise instead of silently degrading."""
    from aidet_amd import RRoIAlign, rbbox_overlaps, rnms
    with pytest.raises(RuntimeError):
        rnms(np.ones((4, 6), np.float32), 0.5)
    with pytest.raises(RuntimeError):
        rbbox_overlaps(torch.ones(2, 5), torch.ones(2, 5))
    with pytest.raises(NotImplementedError):
        RRoIAlign(7, 1.0)(torch.zeros(1, 1, 8, 8), torch.zeros(1, 6))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under aidet_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'aidet_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cuh', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle', src, re.M), f
                assert 'libobb_oracle' not in src, f


def test_module_construction_mirrors_roi_align():
    """RRoIAlign is built like RoIAlign: layer_cls(spatial_scale=1/s, **cfg)
    (single_level.py:45-52) and exposes .out_size (single_level.py:94)."""
    from aidet_amd import RRoIAlign
    cfg = dict(out_size=7, sample_num=2)
    layers = [RRoIAlign(spatial_scale=1 / s, **cfg) for s in (4, 8, 16, 32)]
    assert layers[0].out_size == (7, 7)
    assert 'RRoIAlign(out_size=(7, 7), spatial_scale=0.25, sample_num=2' in repr(layers[0])


def test_torch_extension_loads_and_validates():
    """obbops_torch (the ATen-level binding, aidet_amd/csrc/obbops_torch.cpp): importable
    without a GPU, same version as the C library, reference-style errors."""
    import torch
    from aidet_amd import _lib, torch_ext
    e = torch_ext.ext()
    assert e.version() == _lib.lib().obb_version()
    for name in ('rnms', 'poly_nms', 'rnms_grouped', 'poly_nms_grouped', 'rbbox_overlaps',
                 'rroi_align_forward', 'rroi_align_backward', 'roi_align_forward', 'roi_align_backward', 'nms',
                 'thetaobb2pointobb'):
        assert callable(getattr(e, name))
    # empty input needs no device (nms_wrapper.py:50-51); anything else must live on the GPU
    assert e.rnms(torch.zeros(0, 6), 0.5).dtype == torch.int64
    with pytest.raises(RuntimeError):
        e.rnms(torch.zeros(4, 6), 0.5)          # CPU tensor: C++ check -> RuntimeError
    with pytest.raises(RuntimeError):
        e.rnms(torch.zeros(4, 5), 0.5)          # wrong shape
    with pytest.raises(RuntimeError):
        e.rbbox_overlaps(torch.zeros(2, 5), torch.zeros(2, 5), 'giou', False)


def test_header_is_c99_and_library_links_from_c(tmp_path):
    """include/obbops.h compiles as strict C99 and a C host drives the library (argument
    validation only: no GPU here)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'aidet_amd', 'lib')
    exe = str(tmp_path / 'abi_smoke')
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(root, 'include'),
                           os.path.join(root, 'tests', 'c', 'abi_smoke.c'), '-o', exe, '-L', libdir, '-lobbops',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert 'c abi ok' in out.stdout
