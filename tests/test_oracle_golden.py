"""Pins oracle/ (the C restatement) to the reference's own outputs.

Golden vectors in tests/golden/*.json were produced by the unmodified reference
(tests/golden/gen_golden.py). Bit-exact comparisons throughout.
"""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle as O


def hx(v):
    return float.fromhex(v)


def test_splitmix_and_xorshift_stream(oracle, golden):
    g = golden("rng.json")
    st = C.c_uint64(12374563468)
    got = [oracle.orc_splitmix_next(C.byref(st)) for _ in range(len(g["splitmix_yield_init"]))]
    assert got == [int(v) for v in g["splitmix_yield_init"]]
    assert got[0] == 8946059987183516699
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    got = [oracle.orc_xorshift_next(C.byref(xs)) for _ in range(len(g["xorshift_double"]))]
    assert got == [hx(v) for v in g["xorshift_double"]]
    assert got[0] == 0.40764453281267443


def test_counter_rng_is_random_access_splitmix(oracle):
    # draw #n under a key == (n+1)-th output of a splitmix64 stream seeded with it
    st = C.c_uint64(12374563468)
    seq = [oracle.orc_splitmix_next(C.byref(st)) for _ in range(64)]
    assert [oracle.orc_ctr_key(12374563468, i) for i in range(64)] == seq
    assert oracle.orc_u01(2**64 - 1) == 1.0  # B10: inclusive upper end
    assert oracle.orc_u01(0) == 0.0


def _serial(oracle, case, *, log=False):
    D, pop = case["D"], case["pop"]
    x = np.array(_x0_of(case), dtype=np.float64)
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    lg = None
    if log:
        cap = pop * (case["max_iter"] + 1)
        lx, lf = np.zeros((cap, D)), np.zeros(cap)
        lg = O.EvalLog(lx.ctypes.data_as(O.pd), lf.ctypes.data_as(O.pd), cap, 0, D)
    st = oracle.orc_de_serial(0, 1, 1 if case["strategy"] == "random" else 0,
                              x.ctypes.data_as(O.pd), D, C.byref(xs),
                              hx(case["CR"]), hx(case["F"]), hx(case["eps"]), pop,
                              case["max_iter"], case["no_change"],
                              C.byref(lg) if lg else None)
    after = [oracle.orc_xorshift_next(C.byref(xs)) for _ in range(2)]
    return st, x, after, ((lx, lf, lg.count) if log else None)


_X0 = {"c1_random_pop40_x0_5_7": [5, 7], "random_pop50_x0_5_7": [5, 7],
       "example_best_pop50_x0_2_7": [2, 7]}


def _x0_of(case):
    return case["_x0"]


@pytest.mark.parametrize("name", sorted(_X0))
def test_de_serial_matches_reference_c1(oracle, golden, name):
    case = dict(golden("de_c1.json")[name], _x0=_X0[name])
    st, x, after, _ = _serial(oracle, case)
    assert st.function_calls_used == case["fcalls"]
    assert st.iteration == case["iters"]
    assert st.f_value == hx(case["f"])
    assert x.tolist() == [hx(v) for v in case["x"]]
    assert after == [hx(v) for v in case["rng_after"]]  # generator advanced identically


def test_c1_headline_numbers(golden):
    c = golden("de_c1.json")["c1_random_pop40_x0_5_7"]
    # SURVEY.md §8c G2 / BASELINE.md §2 anchors
    assert (c["fcalls"], c["iters"]) == (1840, 45)
    assert hx(c["f"]) == 5.0733375743553984e-06
    assert [hx(v) for v in c["x"]] == [0.99788203867967407, 0.99569190952076392]
    r = golden("de_c1.json")["readme_objective_pop40"]
    assert (r["fcalls"], r["iters"], hx(r["f"])) == (1840, 45, 9.8894863629873611e-06)


def _fnv(arr):
    h = 1469598103934665603
    for b in np.ascontiguousarray(arr, dtype=np.float64).tobytes():
        h = ((h ^ b) * 1099511628211) & (2**64 - 1)
    return h


_TRACE_X0 = {"pop8_D4": 2.5, "pop40_D2": [5, 7], "pop64_D16": 4.096, "pop256_D128": 4.096}


@pytest.mark.parametrize("strategy", ["random", "best"])
@pytest.mark.parametrize("shape", sorted(_TRACE_X0))
def test_de_serial_matches_reference_traces(oracle, golden, strategy, shape):
    case = golden("de_trace.json")[f"{strategy}_{shape}"]
    x0 = _TRACE_X0[shape]
    case = dict(case, _x0=x0 if isinstance(x0, list) else [x0] * case["D"])
    st, x, after, (lx, lf, n) = _serial(oracle, case, log=True)
    assert n == case["fcalls"] == len(case["eval_f"])
    # every objective value the reference computed, in call order
    assert lf[:n].tolist() == [hx(v) for v in case["eval_f"]]
    # every point it evaluated (init agents, then all trial vectors)
    assert _fnv(lx[:n]) == int(case["eval_x_fnv"])
    if "eval_x" in case:
        ref = np.array([[hx(v) for v in row] for row in case["eval_x"]])
        assert np.array_equal(lx[:n], ref)
    assert x.tolist() == [hx(v) for v in case["x"]]
    assert st.f_value == hx(case["f"])
    assert after == [hx(v) for v in case["rng_after"]]
