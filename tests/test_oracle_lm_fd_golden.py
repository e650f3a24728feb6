"""Pins oracle_lm.c's default-functor LevenbergMarquardt (order = 0: reference arithmetic) to the
reference run with Grad = fin_diff and Hess = fin_diff_h (finite_difference_gradient / _hessian
accuracy 1, nlsolver.h:1385-1517) on built-in objectives."""
import math

import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import hx

OBJ_NAME = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}
CASES = ["rosenbrock_n2_example_start", "rosenbrock_n4_near_minimum", "rosenbrock_n4_indefinite_nan",
         "rosenbrock_n16_6iters", "sphere_n5", "styblinski_tang_n8", "sphere_n64_3iters_lambda1",
         "rosenbrock_n100_2iters", "styblinski_tang_n130_2iters"]  # (past 64 parameters: round 4)


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


def same(a, b):
    return a == b or (math.isnan(a) and math.isnan(b))


@pytest.mark.parametrize("name", CASES)
def test_lm_default_functors_match_reference(oracle, golden, name):
    g = golden("lm_fd.json")[name]
    st, x, lam, flog = O.lm_fd(oracle, OBJ_NAME[g["objective"]], start(g), lam=hx(g["lambda"]),
                               max_iter=g["max_iter"], f_delta=hx(g["f_delta"]), order=0,
                               log_cap=64)
    # 4 n probes per gradient and 16 n^2 per Hessian are all counted function calls (3479-3510)
    assert (st.iteration, st.function_calls_used, st.gradient_evals_used, st.hessian_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"], g["hcalls"])
    assert same(st.f_value, hx(g["f"]))
    assert all(same(a, hx(b)) for a, b in zip(x, g["x"]))
    assert flog[:len(g["f_vals_head"])].tolist() == [hx(v) for v in g["f_vals_head"]]


@pytest.mark.parametrize("name", ["rosenbrock_n4_near_minimum", "sphere_n5", "styblinski_tang_n8"])
def test_tree_order_close_to_reference_arithmetic(oracle, golden, name):
    g = golden("lm_fd.json")[name]
    kw = dict(lam=hx(g["lambda"]), max_iter=2, f_delta=0.0)
    a, xa, _, _ = O.lm_fd(oracle, OBJ_NAME[g["objective"]], start(g), order=1, **kw)
    b, xb, _, _ = O.lm_fd(oracle, OBJ_NAME[g["objective"]], start(g), order=0, **kw)
    assert (a.iteration, a.function_calls_used) == (b.iteration, b.function_calls_used)
    assert abs(a.f_value - b.f_value) <= 1e-6 * max(1.0, abs(b.f_value))
    assert np.max(np.abs(xa - xb)) <= 1e-5 * max(1.0, np.max(np.abs(xb)))
