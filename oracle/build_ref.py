"""Build oracle/_ref/ from the reference's OWN sources (where they lie).

Only meaningful in the build container, where /root/reference exists.  No
reference source is copied into this repository: the compiler reads
/root/reference/mmdet/ops/nms/src/nms_cpu.cpp in place and writes only into
oracle/_ref/ (git-ignored: the compiled module never enters the history).  The
prebuilt oracle/_ref/nms_cpu.so travels with the working-tree snapshot to the GPU
box like the product's own .so files; there it is the live checker of
tests/test_round3_gpu.py::test_hip_nms_and_soft_nms_against_the_reference_build_live
(the HIP `nms` / `soft_nms` against the reference's own compiled loops on fresh
boxes) and of nothing else -- the product never loads it, and a snapshot without
it only skips that test.

Built:  oracle/_ref/nms_cpu.so   -- pybind module of the reference's CPU NMS
        (mmdet/ops/nms/src/nms_cpu.cpp:62-68 `nms`, :212-215 exports).
Not buildable here (stated in DESIGN.md): the .cu kernels (nms_kernel.cu,
roi_align_kernel*.cu) include THC headers that no longer exist in torch 2.x,
and there is no CUDA toolchain.
"""
import os
import sys

REF_SRC = '/root/reference/mmdet/ops/nms/src/nms_cpu.cpp'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')


def build(verbose=False):
    if not os.path.exists(REF_SRC):
        return None
    so = os.path.join(OUT, 'nms_cpu.so')
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(REF_SRC):
        return so
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name='nms_cpu', sources=[REF_SRC], build_directory=OUT,
         extra_cflags=['-O2', '-w'], verbose=verbose)
    return so


def load_ref():
    """Import the prebuilt reference module (None when absent)."""
    so = os.path.join(OUT, 'nms_cpu.so')
    if not os.path.exists(so):
        return None
    import importlib.util
    import torch  # noqa: F401  (the module links against libtorch)
    spec = importlib.util.spec_from_file_location('nms_cpu', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
