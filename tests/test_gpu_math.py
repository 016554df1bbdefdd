"""Device math self-checks (GPU): the shortcuts the accumulation kernel takes must be bit-identical to IEEE."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sigma", [1.0, 0.5, 1.5, 0.75, 2.0])
def test_reciprocal_fma_quotient_is_ieee_exact(sigma):
    """Exhaustive: every float in [1e-27, 1] (about 7.5e8 values) divided by 2*pi*sigma^2 (the kernel only takes the
    shortcut when exp(-dd) > 1e-27, i.e. (h+1)^2/sigma^2 < 60)."""
    from eorb_slam_amd import frontend
    ctx = frontend.Context()
    bad = C.c_uint64(123)
    ctx.check(ctx.L.eorb_selfcheck_division(ctx.h, 1e-27, 1.0, sigma, C.byref(bad)))
    assert bad.value == 0
    ctx.close()


@pytest.mark.parametrize("which,lo,hi", [
    (0, 2.0 ** -30, 104.0),      # exp(-x) for every float x in [2^-30, 104]: 307 232 769 inputs
    (1, 2.0 ** -20, 6.5),        # sin(x), every float in [2^-20, 6.5]: 189 792 257 inputs
    (2, 2.0 ** -20, 6.5),        # cos(x)
    (1, -2.0 ** -20, -6.5),      # the negative half (psi = atan2f(y, x) of KannalaBrandt8::project lies in [-pi, pi])
    (2, -2.0 ** -20, -6.5),
    (3, 2.0 ** -14, 2.35),       # tan(x) as used by KannalaBrandt8::unproject: every float in [2^-14, 2.35] (127 M inputs) ...
    (3, -2.0 ** -14, -2.35),     # ... and their negatives
    (4, 2.0 ** -12, 4096.0),     # atan(x), every float in [2^-12, 4096]
    (4, -2.0 ** -12, -4096.0),
])
def test_device_math_equals_oracle_on_every_input(oracle, which, lo, hi):
    import numpy as np
    from eorb_slam_amd import frontend
    lob = int(np.float32(lo).view(np.uint32)); hib = int(np.float32(hi).view(np.uint32))
    want = oracle.lib(fast=True).orc_math_hash(which, lob, hib)
    ctx = frontend.Context()
    got = C.c_uint64(0)
    ctx.check(ctx.L.eorb_selfcheck_math(ctx.h, which, lob, hib, C.byref(got)))
    ctx.close()
    assert got.value == want


def test_device_atan2f_equals_oracle_on_generated_pairs(oracle):
    """atan2f has two arguments: 2^26 generated (y, x) pairs (every other one raw bit patterns: all exponent combinations, infinities,
    zeros; the rest small rationals as image coordinates give), the same generator on both sides."""
    from eorb_slam_amd import frontend
    L = oracle.lib(fast=True)
    want = L.orc_atan2_hash(0, 64 << 20)
    ctx = frontend.Context()
    got = C.c_uint64(0)
    ctx.check(ctx.L.eorb_selfcheck_math(ctx.h, 5, 0, 63, C.byref(got)))
    ctx.close()
    assert got.value == want
