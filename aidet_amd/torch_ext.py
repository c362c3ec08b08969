"""Loader of ``obbops_torch`` -- the ATen-level (pybind11) binding of the C ABI.

The reference imports its compiled ops as sibling modules of the Python wrappers
(``from . import nms_cpu, nms_cuda``, mmdet/ops/nms/nms_wrapper.py:4;
``from . import roi_align_cuda``, mmdet/ops/roi_align/roi_align.py:6).  This is the
same kind of module for the rotated ops: ``ext().rnms(dets, thr)`` etc. take and return
``torch.Tensor`` (signatures in aidet_amd/csrc/obbops_torch.cpp).  It is optional: the
``ctypes`` path of ``aidet_amd._lib`` serves every wrapper of this package; build this
one with ``aidet_amd.torch_ext.build()`` -- ``torch.utils.cpp_extension.load`` with the
extension's build directory in-tree (``__graft_entry__.build()`` and ``setup.py build_ext`` do).
"""
import importlib.util
import os

import torch  # noqa: F401  (libtorch must be loaded before the extension)

from . import _lib

_EXT = None


def path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'obbops_torch.so')


def build(verbose=False):
    """Compile obbops_torch.so with torch.utils.cpp_extension (host-only C++ against libtorch's
    HIP stream / guard API, linked to libobbops.so next to it; ~40 s, then cached by ninja).
    The reference builds its ops the same way (``CUDAExtension`` in setup.py:90-110)."""
    from torch.utils import cpp_extension
    here = os.path.dirname(os.path.abspath(__file__))
    lib_dir = os.path.join(here, 'lib')
    if not os.path.exists(os.path.join(lib_dir, 'libobbops.so')):
        raise _lib.ObbOpsError('build libobbops.so first (make -C aidet_amd/csrc)')
    src = os.path.join(here, 'csrc', 'obbops_torch.cpp')
    out = path()
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(
            os.path.join(here, '..', 'include', 'obbops.h'))):
        return out
    _lib.lib()   # libobbops.so is loaded before the freshly built module is imported
    cpp_extension.load(
        name='obbops_torch', sources=[src], build_directory=lib_dir, verbose=verbose, is_python_module=True,
        extra_include_paths=[os.path.join(here, '..', 'include'), '/opt/rocm/include'],
        extra_cflags=['-O2', '-D__HIP_PLATFORM_AMD__', '-DUSE_ROCM'],
        extra_ldflags=['-L' + lib_dir, '-lobbops', "-Wl,-rpath,'$$ORIGIN'", '-ltorch_hip', '-lc10_hip'],   # ($$: ninja escape)
        with_cuda=False)
    return out


def ext():
    """The extension module; raises ObbOpsError if it has not been built."""
    global _EXT
    if _EXT is None:
        p = path()
        if not os.path.exists(p):
            raise _lib.ObbOpsError('obbops_torch.so is not built: run aidet_amd.torch_ext.build()')
        _lib.lib()  # libobbops.so first (the extension links to it)
        spec = importlib.util.spec_from_file_location('obbops_torch', p)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _EXT = mod
    return _EXT
