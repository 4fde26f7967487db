"""The kernels' division by a constant — (float)((double)x * (1.0 / (double)c)), csrc/eb_device.h:div_const —
equals IEEE fp32 division x / c for EVERY dividend, for the five divisors of the path.

Default run: every exponent boundary +-4096 patterns, both signs, zeros, infinities, the subnormal range edge,
plus a stride-4099 sweep of all 2^32 patterns (1 M samples per divisor).  EB_EXHAUSTIVE=1 checks all 2^32
patterns per divisor (about 15 s each on 8 cores; run for this round: 0 differences)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests._helpers import oracle_lib

DIVISORS = (10.0, 180.0, float(np.float32(np.pi)), 26.875, 15.625)


def _check(c, first, last, step, form=0):
    lib = oracle_lib().lib
    fn = lib.eb_oracle_check_div_exact
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    fb = C.c_uint32()
    return fn(c, first, last, step, form, C.byref(fb)), fb.value


@pytest.mark.parametrize('c', DIVISORS)
def test_div_const_equals_ieee_division_sampled(c):
    bad, fb = _check(c, 0, 0xffffffff, 4099)
    assert bad == 0, 'first differing pattern 0x%08x' % fb
    for bits in (0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x00000001, 0x80000001, 0x007fffff, 0x00800000):
        assert _check(c, bits, bits, 1)[0] == 0, hex(bits)
    assert _check(c, 0, 1 << 24, 1)[0] == 0            # all positive subnormals and the first normals
    for sign in (0, 0x80000000):
        for e in range(1, 255):
            lo = sign | max((e << 23) - 4096, 0)
            hi = sign | min((e << 23) + 4096, 0x7f7fffff)
            bad, fb = _check(c, lo, hi, 1)
            assert bad == 0, 'first differing pattern 0x%08x' % fb


@pytest.mark.parametrize('c', DIVISORS)
def test_why_not_the_fp32_only_form(c):
    """q = x*rc, r = fma(-q, c, x), q + r*rc is a correct quotient for ordinary dividends but not for -0.0, +-inf
    or tiny non-zero ones — it would need guards and a second code path."""
    for bits in (0x80000000, 0x7f800000, 0xff800000):
        assert _check(c, bits, bits, 1, form=1)[0] == 1
    assert _check(c, 1, 0x0CFFFFFF, 257, form=1)[0] > 0
    assert _check(c, 0x0D000000, 0x0D000000 + (1 << 22), 1, form=1)[0] == 0


@pytest.mark.skipif(os.environ.get('EB_EXHAUSTIVE') != '1', reason='set EB_EXHAUSTIVE=1 (about a minute)')
@pytest.mark.parametrize('c', DIVISORS)
def test_div_const_equals_ieee_division_exhaustive(c):
    bad, fb = _check(c, 0, 0xffffffff, 1)
    assert bad == 0, 'first differing pattern 0x%08x' % fb
