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
