"""The reciprocal-multiply division of the PWA gather / scatter kernels (csrc/pwa.hip `vx_fdivmod`): q = int(float(a) * rb), r = a - q * b, two upward corrections.
The claim the kernels rely on -- for 0 <= a < 2^22 the estimate is the quotient or at most two below it, never above -- is checked here in float32 arithmetic for every
reciprocal the hardware may return (v_rcp_f32 is accurate to 1 ulp: the correctly rounded 1 / b and both of its float32 neighbours), on every dividend for small
divisors and on the dividends around every multiple of b otherwise.  No GPU."""
import numpy as np

SHRINK = np.float32(0.99999976)          # the constant of vx_fd: (1 - 2^-22) in float32


def _fdivmod(a, b, rcp):
    """vx_fdivmod with the reciprocal `rcp` (float32) the hardware returned for b; a: int64 array"""
    rb = np.float32(rcp) * SHRINK
    est = (a.astype(np.float32) * rb).astype(np.int64)        # float -> int conversion truncates, like v_cvt_i32_f32
    q, r = est.copy(), a - est * b
    for _ in range(2):
        up = r >= b
        q, r = q + up, r - b * up
    return q, r, est


def _reciprocals(b):
    r = np.float32(1.0) / np.float32(b)
    return [r, np.nextafter(r, np.float32(0)), np.nextafter(r, np.float32(2))]


def _check(a, b):
    a = np.asarray(a, dtype=np.int64)
    a = a[(a >= 0) & (a < (1 << 22))]
    for rcp in _reciprocals(b):
        q, r, est = _fdivmod(a, b, rcp)
        assert np.array_equal(q, a // b) and np.array_equal(r, a % b), (b, float(rcp))
        d = a // b - est
        assert d.min() >= 0 and d.max() <= 2, (b, float(rcp), int(d.min()), int(d.max()))     # the estimate is never above the quotient, at most two below


def test_every_dividend_for_the_divisors_of_the_shipped_plans():
    a = np.arange(1 << 22, dtype=np.int64)
    for b in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 27, 32, 48, 64, 96, 128, 216, 256, 512, 1000, 4096):
        _check(a, b)


def test_dividends_around_every_multiple_for_all_divisors_up_to_4096_and_some_large_ones():
    rng = np.random.default_rng(7)
    for b in list(range(1, 4097)) + [5000, 32767, 32768, 65535, 65537, 1 << 20, (1 << 22) - 1]:
        k = np.arange(0, (1 << 22) // b + 1, dtype=np.int64)
        if k.size > 4096:
            k = np.concatenate([k[:64], k[-64:], rng.choice(k, 3000, replace=False)])
        a = np.concatenate([k * b - 1, k * b, k * b + 1, k * b + b // 2, rng.integers(0, 1 << 22, 256)])
        _check(a, b)
