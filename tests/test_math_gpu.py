"""The device's deterministic math primitives against the oracle's copies, bit for bit, on
millions of arguments (C-ABI nlsg_probe_math): every engine's bit-exact parity rests on these
(rnorm nlsolver.h:2479-2485 -> det_log / det_cos / det_rnorm; the NLLS model -> det_exp /
det_tanh; Rastrigin test_functions.h:74-76 -> det_cos_2pi), IEEE division and square root
included — det_rnorm writes both without the compiler's scaling and fix-up steps."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu
N = 4_000_000


@pytest.fixture(scope="module")
def probe():
    import torch
    assert torch.cuda.is_available()
    from nlsolver_amd import _capi
    return _capi.probe_math


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def draws(rng, n):
    """64-bit draws: uniform ones, ones with long runs of leading zeros / ones (u1 near 0 and
    near 1), every power of two, and the corners."""
    z = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    sh = rng.integers(0, 64, size=n // 4, dtype=np.uint64)
    z[: n // 4] >>= sh                       # small u1 down to 2^-64
    z[n // 4: n // 2] |= ~(np.uint64(2**64 - 1) >> sh)  # u1 just below 1
    corners = [0, 1, 2, 3, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**64 - 1, 2**64 - 2**11,
               2**64 - 2**10, 2**53, 2**53 + 1] + [2**k for k in range(64)] + \
              [2**64 - 2**k for k in range(64)]
    z[-len(corners):] = np.array(corners, dtype=np.uint64)
    return z


def test_rnorm_bit_exact(probe, oracle):
    z = draws(np.random.default_rng(1), N)
    dev, ref = probe("rnorm", z), O.probe_math(oracle, "rnorm", z)
    bad = np.flatnonzero(dev != ref)
    assert bad.size == 0, (bad[:5], z[bad[:5]], dev[bad[:5]], ref[bad[:5]])
    v = dev.view(np.float64)[N // 2: N - 256]  # the uniform draws
    assert np.isfinite(v).all() and abs(v.mean()) < 0.01 and abs(v.std() - 1.0) < 0.01


def test_u01_bit_exact(probe, oracle):
    z = draws(np.random.default_rng(2), 1_000_000)
    assert np.array_equal(probe("u01", z), O.probe_math(oracle, "u01", z))


@pytest.mark.parametrize("fn,lo,hi", [
    ("log", 0.0, 1.0), ("log", 0.0, 1e300), ("cos", -64.0, 64.0), ("cos", 0.0, 6.2832),
    ("exp", -708.0, 709.0), ("exp", -2.0, 2.0), ("tanh", -25.0, 25.0), ("tanh", -1.0, 1.0),
    ("cos_2pi", -5.12, 5.12), ("cos_2pi", -1e6, 1e6),
])
def test_primitives_bit_exact(probe, oracle, fn, lo, hi):
    rng = np.random.default_rng(hash((fn, lo, hi)) % 2**32)
    x = lo + (hi - lo) * rng.random(N // 4)
    x[: N // 16] *= rng.random(N // 16) ** 8   # crowd towards zero
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2e-308, 1e-310,
                        1.7976931348623157e308, 0.5, 2.0, 64.0, -64.0, 64.00000001, 709.0, 709.1,
                        -708.0, -708.1, 22.0, 22.1, np.pi / 4, np.pi / 2, np.pi, 0.7071067811865476])
    x[-special.size:] = special
    dev, ref = probe(fn, bits(x)), O.probe_math(oracle, fn, bits(x))
    both_nan = np.isnan(dev.view(np.float64)) & np.isnan(ref.view(np.float64))
    bad = np.flatnonzero((dev != ref) & ~both_nan)
    assert bad.size == 0, (fn, x[bad[:5]], dev[bad[:5]], ref[bad[:5]])
