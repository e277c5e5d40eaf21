"""The division shortcuts of the kernels (rc_device.h div_log2_, div_safe_, div_const_) must give
the bits of IEEE division.  CPU part: the log2 shortcut for every mantissa and every reciprocal
seed within 1 ulp (numpy emulation of the fused steps in float64, exact for these magnitudes).
GPU part: the library's self-test on the device's real v_rcp_f32."""
import numpy as np
import pytest


def _fma(a, b, c):
    # float32 fma through float64: the product of two float32 is exact in float64 and the sum of
    # that with a float32 of comparable magnitude rounds once more only below float32 precision
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_log2_division_shortcut_all_mantissas():
    f = np.float32
    bad = 0
    for lo in range(0, 1 << 23, 1 << 20):
        m = (np.arange(lo, lo + (1 << 20), dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
        n, d = m - f(1), m + f(1)
        want = n / d
        r0 = f(1) / d
        for step in (-1, 0, 1):
            r = (r0.view(np.int32) + step).view(np.float32)
            r = _fma(_fma(-d, r, np.ones_like(r)), r, r)
            q = n * r
            q = _fma(_fma(-d, q, n), r, q)
            bad += int((q.view(np.uint32) != want.view(np.uint32)).sum())
    assert bad == 0


@pytest.mark.gpu
def test_device_selftest(rc_lib):
    from retrocapture_amd import engine
    assert engine.selftest_fastmath(0) == [0, 0, 0]
