"""Pins oracle/oracle_bfgs.c (tree=0: reference arithmetic) to the reference's BFGS."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import _fnv, hx

CASES = ["n8", "n64", "n1024", "n64_default_stop", "n100_ragged_start", "n130_alpha_half",
         "n256_max_iter_5"]


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


@pytest.mark.parametrize("name", CASES)
def test_bfgs_serial_matches_reference(oracle, golden, name):
    g = golden("bfgs.json")[name]
    st, x, flog = O.bfgs_quad(oracle, start(g), max_iter=g["max_iter"], grad_eps=hx(g["grad_eps"]),
                              alpha=hx(g["alpha"]), tree=0, log=True)
    assert (st.iteration, st.function_calls_used, st.gradient_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"])
    assert st.f_value == hx(g["f"])
    assert _fnv(x) == int(g["x_fnv"])
    assert x[:8].tolist() == [hx(v) for v in g["x_head"]]
    # every objective value of the run, in call order (line-search trials included)
    assert flog.tolist() == [hx(v) for v in g["f_vals"]]


def test_update_inverse_hessian_matches_reference(oracle, golden):
    g = golden("bfgs.json")["hess_update_3x3"]
    H = np.array([2.0, 0.3, -0.1, 0.3, 1.5, 0.2, -0.1, 0.2, 1.1])
    s, y, t = np.array([0.3, -0.2, 0.5]), np.array([0.7, 0.1, -0.4]), np.zeros(3)
    rho = hx(g["rho"])
    assert rho == 1.0 / (y[0] * s[0] + y[1] * s[1] + y[2] * s[2])
    oracle.orc_update_inverse_hessian(O._ptr(H), O._ptr(s), O._ptr(y), O._ptr(t), rho, 3, 0)
    assert H.tolist() == [hx(v) for v in g["H"]] and t.tolist() == [hx(v) for v in g["t"]]


@pytest.mark.parametrize("name", CASES)
def test_tree_order_agrees_with_reference_arithmetic(oracle, golden, name):
    """The kernel's summation tree vs the reference's sequential sums: same iterations,
    objective within 1e-12 relative (north_star tolerance)."""
    g = golden("bfgs.json")[name]
    st, x, _ = O.bfgs_quad(oracle, start(g), max_iter=g["max_iter"], grad_eps=hx(g["grad_eps"]),
                           alpha=hx(g["alpha"]), tree=1)
    ref, xr, _ = O.bfgs_quad(oracle, start(g), max_iter=g["max_iter"], grad_eps=hx(g["grad_eps"]),
                             alpha=hx(g["alpha"]), tree=0)
    assert abs(st.f_value - ref.f_value) <= 1e-12 * abs(ref.f_value)
    if hx(g["grad_eps"]) >= 1e-6 or g["max_iter"] < 10:
        # a stop test well above rounding noise: identical work
        assert (st.iteration, st.function_calls_used) == (ref.iteration, ref.function_calls_used)
    else:
        # grad_eps = 1e-10 stops on |norm_k - norm_{k-1}| at the rounding-noise floor:
        # the count may move by a few iterations, the optimum may not
        assert abs(int(st.iteration) - int(ref.iteration)) <= 6
    assert np.max(np.abs(x - xr)) <= 1e-6


def test_survey_anchor(golden):
    g = golden("bfgs.json")
    assert abs(hx(g["n64"]["f"]) - (-3.551897415725755)) < 1e-13
    assert abs(hx(g["n1024"]["f"]) - (-65.308652606191941)) < 1e-10


@pytest.mark.parametrize("name", ["n8", "n64", "n1024", "n64_default_stop"])
def test_symmetric_restatement_within_1e12_of_reference(oracle, golden, name):
    """tree = 2 (update with denom * (s[i] s[j]), products summed block-wise as the symmetric
    kernels stream the upper blocks of H) against the reference's own runs: f within 1e-12; the
    inverse Hessian it carries is bitwise symmetric at every iteration."""
    from tests import _oracle as O
    c = golden("bfgs.json")[name]
    n = c["n"]
    x0 = np.full(n, float.fromhex(c["x0"]))
    kw = dict(max_iter=c["max_iter"], grad_eps=float.fromhex(c["grad_eps"]),
              alpha=float.fromhex(c["alpha"]))
    st, x, _ = O.bfgs_quad(oracle, x0, tree=2, **kw)
    fref = float.fromhex(c["f"])
    assert abs(st.f_value - fref) <= 1e-12 * abs(fref)
    if kw["grad_eps"] >= 1e-6:
        assert (st.iteration, st.function_calls_used, st.gradient_evals_used) == \
            (c["iters"], c["fcalls"], c["gcalls"])
    for k in (1, 2, 5, 12):
        H = np.zeros((n, n))
        O.bfgs_quad(oracle, x0, tree=2, hessian=H, max_iter=k, grad_eps=0.0, alpha=kw["alpha"])
        assert np.array_equal(H, H.T) and not np.isnan(H).any()
