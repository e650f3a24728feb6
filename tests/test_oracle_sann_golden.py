"""Pins oracle_sann.c's serial SANN (reference arithmetic, xorshift draw order) to runs of the
reference's SANN class (nlsolver.h:2744-2815), and relates the synchronous variant (what the GPU
executes) to it."""
import math

import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import hx

OBJ_NAME = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}
CASES = ["rosenbrock_n2_default_schedule", "rosenbrock_n8", "sphere_n16_hot", "styblinski_tang_n6",
         "sphere_n4_maximize", "rosenbrock_n130_ragged", "temp_iter_1_no_moves"]


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


@pytest.mark.parametrize("name", CASES)
def test_sann_serial_matches_reference(oracle, golden, name):
    g = golden("sann.json")[name]
    st, x, nxt, flog = O.sann_serial(oracle, OBJ_NAME[g["objective"]], start(g),
                                     minimize=bool(g["minimize"]), max_iter=g["max_iter"],
                                     temp_iter=g["temp_iter"], temp_max=hx(g["temp_max"]),
                                     log_cap=64)
    assert (st.iteration, st.function_calls_used) == (g["iters"], g["fcalls"])
    assert st.f_value == hx(g["f"])
    assert np.array_equal(x, np.array([hx(v) for v in g["x"]]))
    assert nxt == hx(g["next_draw"])  # the same number of draws was consumed
    head = [hx(v) for v in g["f_vals_head"]]
    assert flog[:len(head)].tolist() == head


def test_sann_counts_follow_the_schedule(oracle):
    """1 + max_iter * (temperature_iter - 1) evaluations (:2781, 2795-2803)."""
    for temp_iter in (1, 2, 10):
        st, _, _ = O.sann_sync(oracle, "sphere", np.ones(5), 7, 0, max_iter=30, temp_iter=temp_iter)
        assert st.iteration == 30 and st.function_calls_used == 1 + 30 * (temp_iter - 1)


def test_sann_sync_is_a_valid_chain(oracle):
    """The synchronous variant keeps the chain's invariants: the reported best is the objective at
    the returned point and never worse than the start; chains with different keys differ."""
    x0 = np.full(8, 0.5) + 0.1 * np.arange(8)
    f0 = oracle.orc_objective_tree(0, x0.ctypes.data_as(O.pd), 8)
    seen = set()
    for chain in range(4):
        st, x, flog = O.sann_sync(oracle, "rosenbrock", x0, 12374563468, chain, max_iter=100,
                                  log_cap=901)
        assert st.f_value == oracle.orc_objective_tree(0, x.ctypes.data_as(O.pd), 8)
        assert st.f_value <= f0 and st.f_value == min(flog.min(), f0)
        seen.add(st.f_value)
    assert len(seen) == 4


def test_sann_sync_maximize_negates(oracle):
    x0 = np.array([1.0, 1.5, 2.0, 2.5])
    st, x, _ = O.sann_sync(oracle, "sphere", x0, 5, 1, minimize=False, max_iter=50)
    assert st.f_value == -oracle.orc_objective_tree(1, x.ctypes.data_as(O.pd), 4)
    assert -st.f_value >= float(x0 @ x0)
    assert not math.isnan(st.f_value)
