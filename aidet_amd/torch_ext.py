"""Loader of ``obbops_torch`` -- the ATen-level (pybind11) binding of the C ABI.

The reference imports its compiled ops as sibling modules of the Python wrappers
(``from . import nms_cpu, nms_cuda``, mmdet/ops/nms/nms_wrapper.py:4;
``from . import roi_align_cuda``, mmdet/ops/roi_align/roi_align.py:6).  This is the
same kind of module for the rotated ops: ``ext().rnms(dets, thr)`` etc. take and return
``torch.Tensor`` (signatures in aidet_amd/csrc/obbops_torch.cpp).  It is optional: the
``ctypes`` path of ``aidet_amd._lib`` serves every wrapper of this package; build this
one with ``make -C aidet_amd/csrc torch`` (``__graft_entry__.build()`` does).
"""
import importlib.util
import os

import torch  # noqa: F401  (libtorch must be loaded before the extension)

from . import _lib

_EXT = None


def path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'obbops_torch.so')


def ext():
    """The extension module; raises ObbOpsError if it has not been built."""
    global _EXT
    if _EXT is None:
        p = path()
        if not os.path.exists(p):
            raise _lib.ObbOpsError('obbops_torch.so is not built: run `make -C aidet_amd/csrc torch`')
        _lib.lib()  # libobbops.so first (the extension links to it)
        spec = importlib.util.spec_from_file_location('obbops_torch', p)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _EXT = mod
    return _EXT
