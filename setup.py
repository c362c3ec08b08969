"""Build hook: `python setup.py build_ext --inplace` (or `pip install -e .` offline) compiles
libobbops.so (HIP kernels + C ABI, gfx950; aidet_amd/csrc/Makefile) and obbops_torch.so (ATen binding;
torch.utils.cpp_extension, aidet_amd/torch_ext.py) in-tree -- the counterpart of the reference's `make_cuda_ext(...)` /
`CUDAExtension` entries and `package_data={'mmdet.ops': ['*/*.so']}` (setup.py:90-110,204,224-298).
"""
import os
import subprocess
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class build_ext(Command):
    description = 'compile libobbops.so and obbops_torch.so with hipcc / g++'
    user_options = [('inplace', 'i', 'ignored: the libraries are always built in-tree'),
                    ('no-torch', None, 'skip the ATen binding')]
    boolean_options = ['inplace', 'no-torch']

    def initialize_options(self):
        self.inplace = 0
        self.no_torch = 0

    def finalize_options(self):
        pass

    def run(self):
        csrc = os.path.join(HERE, 'aidet_amd', 'csrc')
        subprocess.check_call(['make', '-C', csrc, '-j4'])
        if not self.no_torch:
            sys.path.insert(0, HERE)
            from aidet_amd import torch_ext
            torch_ext.build()


class build_py_with_libs(build_py):
    def run(self):
        self.run_command('build_ext')
        build_py.run(self)


setup(
    name='aidet_amd',
    version='1.1.0',
    description='MI355X-native oriented-bounding-box operators (rotated IoU / NMS / RoI-Align) for AIDet',
    packages=find_packages(include=['aidet_amd', 'aidet_amd.*']),
    package_data={'aidet_amd': ['lib/*.so']},
    cmdclass={'build_ext': build_ext, 'build_py': build_py_with_libs},
    python_requires='>=3.8',
)
