"""Pins oracle_nmpso.c's serial NelderMeadPSO (reference arithmetic, xorshift draw order) to runs
of the reference class (nlsolver.h:3546-3920, unbounded overloads), and relates the synchronous
variant (what the GPU executes) to it."""
import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import hx

OBJ_NAME = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}
CASES = ["rosenbrock_n2_defaults", "rosenbrock_n4", "rosenbrock_n8_200iters", "rosenbrock_n16",
         "sphere_n6", "styblinski_tang_n4_maximize", "rosenbrock_n130_ragged"]


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


@pytest.mark.parametrize("name", CASES)
def test_nmpso_serial_matches_reference(oracle, golden, name):
    g = golden("nmpso.json")[name]
    st, x, nxt, flog = O.nmpso_serial(oracle, OBJ_NAME[g["objective"]], start(g),
                                      minimize=bool(g["minimize"]), eps=hx(g["eps"]),
                                      max_iter=g["max_iter"], no_change=g["no_change"], log_cap=64)
    assert (st.iteration, st.function_calls_used) == (g["iters"], g["fcalls"])
    assert st.f_value == hx(g["f"])
    assert np.array_equal(x, np.array([hx(v) for v in g["x"]]))
    assert nxt == hx(g["next_draw"])  # the same number of draws was consumed
    head = [hx(v) for v in g["f_vals_head"]]
    assert flog[:len(head)].tolist() == head


def test_nmpso_one_dimension_is_refused(oracle):
    st, x, _, _ = O.nmpso_serial(oracle, "sphere", np.array([2.0]))
    assert (st.f_value, st.iteration, st.function_calls_used) == (999999, 0, 0) and x[0] == 2.0


def test_nmpso_sync_is_a_valid_run(oracle):
    x0 = 0.5 + 0.1 * np.arange(6)
    f0 = oracle.orc_objective_tree(0, x0.ctypes.data_as(O.pd), 6)
    seen = set()
    for inst in range(3):
        st, x, flog = O.nmpso_sync(oracle, "rosenbrock", x0, 12374563468, inst, eps=0.0,
                                   max_iter=80, no_change=1000, log_cap=4000)
        assert st.iteration == 80
        assert st.f_value == oracle.orc_objective_tree(0, x.ctypes.data_as(O.pd), 6)
        assert st.f_value <= f0
        assert st.function_calls_used <= 4000 and st.f_value == flog[:st.function_calls_used].min()
        seen.add(st.f_value)
    assert len(seen) == 3


def test_nmpso_sync_bounded_stays_inside(oracle):
    x0 = np.array([0.5, -0.3, 0.8, 0.1])
    st, x, _ = O.nmpso_sync(oracle, "styblinski_tang", x0, 3, 0, upper=1.0, lower=-1.0, eps=0.0,
                            max_iter=60, no_change=1000)
    assert st.f_value == oracle.orc_objective_tree(2, x.ctypes.data_as(O.pd), 4)
