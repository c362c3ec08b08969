"""The gaussian Soft-NMS weight of the reference is std::exp on float = glibc's expf (nms_cpu.cpp:176-178).  The kernels
restate that routine (aidet_amd/csrc/obb_softnms.hip, soft_exp: exp(x) = 2^(k/32) * 2^(r/32), a table of 32 doubles and a
cubic in double, one rounding to float).  This test pins the restated ALGORITHM -- the same table rule and constants,
written in numpy -- against the libm of this image bit for bit; the GPU tests then pin the kernels against the oracle's
loop, which calls that libm (tests/test_round4_gpu.py, tests/test_gpu_parity.py: gaussian arrays, array_equal)."""
import ctypes
import re
import os
import struct
from decimal import Decimal, getcontext
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 32


def _table():
    return np.array([int(np.float64(2.0 ** (i / N)).view(np.uint64)) - ((i << 52) // N) for i in range(N)], dtype=np.uint64)


def restated_expf(x):
    x = np.asarray(x, dtype=np.float32)
    c0 = float.fromhex('0x1.c6af84b912394p-5') / N / N / N
    c1 = float.fromhex('0x1.ebfce50fac4f3p-3') / N / N
    c2 = float.fromhex('0x1.62e42ff0c52d6p-1') / N
    inv_ln2_n = float.fromhex('0x1.71547652b82fep+0') * N
    shift = float.fromhex('0x1.8p+52')
    z = inv_ln2_n * x.astype(np.float64)
    kd = z + shift
    ki = kd.view(np.uint64)
    r = z - (kd - shift)
    s = (_table()[(ki & np.uint64(N - 1)).astype(np.int64)] + (ki << np.uint64(47))).view(np.float64)
    y = ((c0 * r + c1) * (r * r) + (c2 * r + 1.0)) * s
    out = y.astype(np.float32)
    out[x < np.float32(float.fromhex('-0x1.9fe368p6'))] = 0.0
    return out


def test_restated_expf_is_the_libm_of_this_image():
    libm = ctypes.CDLL('libm.so.6')
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(5)
    xs = [-(rng.random(60000) ** 2 * 4).astype(np.float32),            # -ovr^2 / sigma for sigma >= 0.25
          -(rng.random(20000) * 110).astype(np.float32),               # down to the underflow
          np.float32([-0.0, -1e-30, -1e-10, -87.3, -88.0, -103.9, -104.0, -200.0])]
    for x in xs:
        want = np.array([libm.expf(float(v)) for v in x], dtype=np.float32)
        got = restated_expf(x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_kernel_source_holds_the_same_table_and_constants():
    src = open(os.path.join(ROOT, 'aidet_amd', 'csrc', 'obb_softnms.hip')).read()
    body = src[src.index('kExpfTab[32]'):]
    words = [int(w, 16) for w in re.findall(r'0x([0-9a-f]{16})ull', body[:body.index('};')])]
    assert words == [int(v) for v in _table()]
    for c in ('0x1.c6af84b912394p-5', '0x1.ebfce50fac4f3p-3', '0x1.62e42ff0c52d6p-1', '0x1.71547652b82fep+0', '0x1.8p+52', '0x1.9fe368p6f'):
        assert c in body[:4000], c


# ---- float64 rows: glibc's exp (double), in the form an x86-64 host with fused multiply-add runs it -------------------
def _bits(d):
    return struct.unpack('<Q', struct.pack('<d', d))[0]


def _dbl(b):
    return struct.unpack('<d', struct.pack('<Q', b & (2 ** 64 - 1)))[0]


def _table64():
    getcontext().prec = 90
    out = []
    for k in range(128):
        e = Decimal(2) ** (Decimal(k) / Decimal(128))
        h = float(e)
        tail = float((Fraction(e) - Fraction(h)) / Fraction(h))
        out += [_bits(tail), _bits(h) - ((k << 52) // 128)]
    return out


def _fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))       # one rounding


def restated_exp(x, table):
    inv_ln2_n = float.fromhex('0x1.71547652b82fep0') * 128
    hi, lo = float.fromhex('-0x1.62e42fefa0000p-8'), float.fromhex('-0x1.cf79abc9e3b3ap-47')
    shift = float.fromhex('0x1.8p52')
    c2, c3, c4, c5 = (float.fromhex(h) for h in ('0x1.ffffffffffdbdp-2', '0x1.555555555543cp-3', '0x1.55555cf172b91p-5', '0x1.1111167a4d017p-7'))
    kd = _fma(inv_ln2_n, x, shift)
    ki = _bits(kd)
    kd -= shift
    r = _fma(kd, lo, _fma(kd, hi, x))
    idx = 2 * (ki % 128)
    tail, scale = _dbl(table[idx]), _dbl(table[idx + 1] + (ki << 45))
    r2 = r * r
    s1 = _fma(r2, _fma(r, c3, c2), tail + r)
    tmp = _fma(r2 * r2, _fma(r, c5, c4), s1)
    return _fma(scale, tmp, scale)


def _host_has_fma():
    try:
        return ' fma ' in open('/proc/cpuinfo').read()
    except OSError:
        return False


def test_restated_double_exp_is_the_libm_of_this_image():
    import pytest
    if not _host_has_fma():
        pytest.skip('glibc runs its uncontracted exp on this host: another last bit for 0.07 % of the arguments')
    libm = ctypes.CDLL('libm.so.6')
    libm.exp.restype = ctypes.c_double
    libm.exp.argtypes = [ctypes.c_double]
    table = _table64()
    rng = np.random.default_rng(11)
    xs = np.concatenate([-(rng.random(3000) ** 2 * 4), -rng.random(1000) * 500, [-1e-17, -1e-10, -0.5, -511.9]])
    for x in xs:
        assert _bits(restated_exp(float(x), table)) == _bits(libm.exp(float(x))), x


def test_kernel_source_holds_the_double_table_and_constants():
    src = open(os.path.join(ROOT, 'aidet_amd', 'csrc', 'obb_softnms.hip')).read()
    body = src[src.index('kExpTab[256]'):]
    words = [int(w, 16) for w in re.findall(r'0x([0-9a-f]{16})ull', body[:body.index('};')])]
    assert words == _table64()
    for c in ('0x1.71547652b82fep0', '-0x1.62e42fefa0000p-8', '-0x1.cf79abc9e3b3ap-47', '0x1.ffffffffffdbdp-2', '0x1.555555555543cp-3',
              '0x1.55555cf172b91p-5', '0x1.1111167a4d017p-7'):
        assert c in body[:12000], c
