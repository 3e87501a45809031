"""The NVF mask's last operation, variance / (1 + variance) (nvf.hpp:50), is formed by the kernels with a 4-operation
sequence (reciprocal, product, one residual correction) instead of the IEEE division sequence (csrc/wm_device.hpp nvf_quot).  "Bit-exact against the oracle" then rests on
that sequence giving the correctly rounded quotient for EVERY variance the mask can produce -- a one-parameter family (the
divisor is 1 + the dividend), so it is checked exhaustively on the device: every f32 in [-0.5, 2^17)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

# variance = sumSq / p^2 - mean^2 of pixels in [0, 255]: at most 255^2 = 65025 < 2^17; rounding can take it below zero by
# far less than 0.5 (a constant region: |error| <~ 0.1)
# (-0.0 itself is excluded: q - mean^2 is never -0 under round-to-nearest, and the sequence returns +0 for it)
RANGES = [(0x00000000, 0x48000000), (0x80000001, 0xBF000000)]


def run(wm, variant, lo, hi):
    bad, first = C.c_ulonglong(0), C.c_uint32(0)
    rc = wm.lib().wm_selftest_nvf_quotient(0, variant, lo, hi, C.byref(bad), C.byref(first))
    assert rc == 0, wm.strerror(rc)
    return bad.value, first.value


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_nvf_quotient_exhaustive(wm, variant):
    """variant 2 = what the kernels use, 1 / 0 = the 6- and 8-operation sequences with a refined reciprocal (earlier rounds):
    no differing value in the domain"""
    import torch
    assert torch.cuda.is_available()
    total = 0
    for lo, hi in RANGES:
        bad, first = run(wm, variant, lo, hi)
        assert bad == 0, f"variant {variant}: {bad} of {hi - lo} values differ from the IEEE quotient, first at bits {first:#010x}"
        total += hi - lo
    assert total > 2_000_000_000


def test_nvf_quotient_selftest_can_fail(wm):
    """the product with the hardware reciprocal alone is NOT exact: the self-test must say so (it is not vacuous)"""
    bad = sum(run(wm, 3, lo, hi)[0] for lo, hi in RANGES)
    assert bad > 1000
