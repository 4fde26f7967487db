"""The kernels' 3-op division by a constant (csrc/eb_device.h:div_fast) equals IEEE fp32 division, except for
the dividends the kernels send to the true division: non-zero |x| < 2^-101, -0.0, +-inf.

Default run: every exponent boundary +-4096 patterns, both signs, plus a stride-4099 sweep of all 2^32
patterns (1 M samples per divisor).  EB_EXHAUSTIVE=1 checks all 2^32 patterns per divisor (a minute or two on
8 cores; run for this round: 0 differences for the five divisors)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests._helpers import oracle_lib

DIVISORS = (10.0, 180.0, float(np.float32(np.pi)), 26.875, 15.625)


def _check(c, first, last, step, guard=1):
    lib = oracle_lib().lib
    fn = lib.eb_oracle_check_div_exact
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    fb = C.c_uint32()
    return fn(c, first, last, step, guard, C.byref(fb)), fb.value


@pytest.mark.parametrize('c', DIVISORS)
def test_the_guarded_dividends_are_the_ones_that_need_it(c):
    """unguarded: -0.0 and +-inf differ, and so do some tiny magnitudes — but none at or above 2^-101"""
    for bits in (0x80000000, 0x7f800000, 0xff800000):
        assert _check(c, bits, bits, 1, guard=0)[0] == 1
    assert _check(c, 1, 0x0CFFFFFF, 1, guard=0)[0] > 0
    assert _check(c, 0x0D000000, 0x0D000000 + (1 << 24), 1, guard=0)[0] == 0


@pytest.mark.parametrize('c', DIVISORS)
def test_div_fast_equals_ieee_division_sampled(c):
    bad, fb = _check(c, 0, 0xffffffff, 4099)
    assert bad == 0, 'first differing pattern 0x%08x' % fb
    for sign in (0, 0x80000000):
        for e in range(1, 255):
            lo = sign | max((e << 23) - 4096, 0)
            hi = sign | min((e << 23) + 4096, 0x7f7fffff)
            bad, fb = _check(c, lo, hi, 1)
            assert bad == 0, 'first differing pattern 0x%08x' % fb


@pytest.mark.skipif(os.environ.get('EB_EXHAUSTIVE') != '1', reason='set EB_EXHAUSTIVE=1 (minutes)')
@pytest.mark.parametrize('c', DIVISORS)
def test_div_fast_equals_ieee_division_exhaustive(c):
    bad, fb = _check(c, 0, 0xffffffff, 1)
    assert bad == 0, 'first differing pattern 0x%08x' % fb
