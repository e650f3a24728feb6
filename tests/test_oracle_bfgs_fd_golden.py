"""Pins oracle_bfgs.c's finite-difference model (tree = 0: reference arithmetic) to the
reference's BFGS run with its DEFAULT gradient, fin_diff = finite_difference_gradient<.,.,1>
(nlsolver.h:1385-1413, 2849-2855, 3218-3224) on built-in objectives."""
import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import hx

OBJ_NAME = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}
CASES = ["rosenbrock_n2", "rosenbrock_n4", "rosenbrock_n16_default_stop", "rosenbrock_n128_20iters",
         "sphere_n5", "sphere_n130_alpha_half", "styblinski_tang_n8"]


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


@pytest.mark.parametrize("name", CASES)
def test_bfgs_finite_difference_matches_reference(oracle, golden, name):
    g = golden("bfgs_fd.json")[name]
    st, x, flog, count = O.bfgs_fd(oracle, OBJ_NAME[g["objective"]], start(g),
                                   max_iter=g["max_iter"], grad_eps=hx(g["grad_eps"]),
                                   alpha=hx(g["alpha"]), tree=0, log_cap=64)
    # 4 n probes per gradient all count as function calls (the counting wrapper, 3218-3224)
    assert (st.iteration, st.function_calls_used, st.gradient_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"])
    assert count == g["f_count"] == g["fcalls"]
    assert st.f_value == hx(g["f"])
    assert x.tolist() == [hx(v) for v in g["x"]]
    assert flog.tolist() == [hx(v) for v in g["f_vals_head"]]


@pytest.mark.parametrize("name", ["rosenbrock_n4", "sphere_n5", "styblinski_tang_n8",
                                  "rosenbrock_n128_20iters"])
def test_tree_order_close_to_reference_arithmetic(oracle, golden, name):
    """The kernel's lane tree vs sequential sums. Finite differences divide rounding noise by
    12 * 2.2e-8, so trajectories drift apart sooner than with an analytic gradient: compare a
    short horizon tightly and the end point loosely."""
    g = golden("bfgs_fd.json")[name]
    kw = dict(grad_eps=hx(g["grad_eps"]), alpha=hx(g["alpha"]))
    obj = OBJ_NAME[g["objective"]]
    a, xa, _, _ = O.bfgs_fd(oracle, obj, start(g), max_iter=3, tree=1, **kw)
    b, xb, _, _ = O.bfgs_fd(oracle, obj, start(g), max_iter=3, tree=0, **kw)
    if obj == "rosenbrock":  # far from converged after 3 iterations: identical work
        assert (a.iteration, a.gradient_evals_used) == (b.iteration, b.gradient_evals_used)
    # (the convex ones reach the rounding floor within 2-3 iterations, where the stop test
    # |norm_k - norm_{k-1}| < grad_eps may fire one iteration apart)
    assert abs(a.f_value - b.f_value) <= 1e-6 * max(1.0, abs(b.f_value))
    assert np.max(np.abs(xa - xb)) <= 1e-5 * max(1.0, np.max(np.abs(xb)))
