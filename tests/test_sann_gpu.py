"""Parity tests: batched HIP simulated annealing (one wave per chain) vs oracle_sann.c's
synchronous variant (counter-keyed draws, deterministic log / cos / exp, objective tree):
bit-exact best points, values and counters for every chain."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu
SEED = 12374563468


@pytest.fixture(scope="module")
def mod():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd


def starts(batch, n, base, spread, seed=0):
    rng = np.random.default_rng(seed + n)
    return base + spread * (rng.random((batch, n)) - 0.5)


@pytest.mark.parametrize("objective,n,kw", [
    ("rosenbrock", 2, dict(max_iter=200, temperature_iter=10, temperature_max=10.0)),
    ("rosenbrock", 8, dict(max_iter=120, temperature_iter=10, temperature_max=10.0)),
    ("sphere", 16, dict(max_iter=60, temperature_iter=5, temperature_max=50.0)),
    ("styblinski_tang", 6, dict(max_iter=150, temperature_iter=10, temperature_max=10.0)),
    ("rosenbrock", 127, dict(max_iter=30, temperature_iter=4, temperature_max=10.0)),
    ("rosenbrock", 128, dict(max_iter=30, temperature_iter=10, temperature_max=10.0)),
    ("rosenbrock", 130, dict(max_iter=20, temperature_iter=10, temperature_max=2.0)),
    ("sphere", 300, dict(max_iter=15, temperature_iter=6, temperature_max=10.0)),
    ("styblinski_tang", 1000, dict(max_iter=8, temperature_iter=5, temperature_max=10.0)),
    ("rosenbrock", 1, dict(max_iter=10, temperature_iter=10, temperature_max=10.0)),
])
@pytest.mark.parametrize("minimize", [True, False])
def test_sann_chains_bit_exact_vs_sync_oracle(mod, oracle, objective, n, kw, minimize):
    batch = 7
    x0 = starts(batch, n, 0.5, 1.0)
    with mod.SANNEngine(objective, batch, n, minimize=minimize, seed=SEED, chain_lo=3, **kw) as eng:
        x, st = eng.minimize(x0)
    for b in range(batch):
        ref, xr, _ = O.sann_sync(oracle, objective, x0[b], SEED, 3 + b, minimize=minimize,
                                 max_iter=kw["max_iter"], temp_iter=kw["temperature_iter"],
                                 temp_max=kw["temperature_max"])
        assert np.array_equal(x[b], xr), (b, x[b], xr)
        assert st[b].f_value == ref.f_value
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)


@pytest.mark.parametrize("temperature_iter,max_iter", [(0, 5), (1, 25), (2, 0), (10, 0)])
def test_sann_degenerate_schedules(mod, oracle, temperature_iter, max_iter):
    """temperature_iter <= 1: no trial points at all (for j = 1; j < temperature_iter, :2795);
    max_iter = 0: only the start is scored (:2781, 2788)."""
    x0 = starts(3, 4, 0.3, 0.4)
    with mod.SANNEngine("rosenbrock", 3, 4, max_iter=max_iter,
                        temperature_iter=temperature_iter) as eng:
        x, st = eng.minimize(x0)
    assert np.array_equal(x, x0)
    for b in range(3):
        assert st[b].function_calls_used == 1 and st[b].iteration == max_iter
        assert st[b].f_value == oracle.orc_objective_tree(0, x0[b].ctypes.data_as(O.pd), 4)


def test_sann_long_schedule_is_cut_into_launches(mod, oracle):
    """A schedule longer than one launch's span (65 536 trial points): the chain's state crosses
    the launches through HBM and the result equals the single-pass oracle."""
    x0 = starts(2, 4, 0.5, 1.0)
    kw = dict(max_iter=9000, temperature_iter=10, temperature_max=10.0)
    with mod.SANNEngine("rosenbrock", 2, 4, seed=SEED, **kw) as eng:
        x, st = eng.minimize(x0)
    for b in range(2):
        ref, xr, _ = O.sann_sync(oracle, "rosenbrock", x0[b], SEED, b, max_iter=9000, temp_iter=10,
                                 temp_max=10.0)
        assert np.array_equal(x[b], xr) and st[b].f_value == ref.f_value
        assert st[b].function_calls_used == ref.function_calls_used == 1 + 9000 * 9


def test_sann_chain_ids_are_global(mod):
    """chain_lo shifts the keys: chains [2, 5) of one engine = chains [0, 3) of an engine created
    with chain_lo = 2 (how a batch is split across ranks)."""
    x0 = starts(5, 10, 0.5, 1.0)
    kw = dict(max_iter=40, seed=99)
    with mod.SANNEngine("rosenbrock", 5, 10, **kw) as eng:
        xa, sa = eng.minimize(x0)
    with mod.SANNEngine("rosenbrock", 3, 10, chain_lo=2, **kw) as eng:
        xb, sb = eng.minimize(x0[2:])
    assert np.array_equal(xa[2:], xb)
    assert [s.f_value for s in sa[2:]] == [s.f_value for s in sb]


def test_sann_finds_the_sphere_minimum_region(mod):
    """Behavioural check in the spirit of the reference's tests (within tolerance of the known
    minimum on a small problem): the best of 64 chains on Sphere-2D."""
    x0 = np.full((64, 2), 3.0)
    solver = mod.SANN("sphere", None, 2000, 10, 10.0)
    st = solver.minimize(x0)
    best = min(s.f_value for s in st)
    assert best < 0.05, best
    assert all(s.f_value <= 18.0 for s in st)  # never worse than the start


def test_sann_drop_in_class_single_start(mod, oracle):
    x = np.array([2.0, 7.0])
    st = mod.SANN("rosenbrock", None, 300).minimize(x)
    ref, xr, _ = O.sann_sync(oracle, "rosenbrock", np.array([2.0, 7.0]),
                             mod.de.DEFAULT_SEED, 0, max_iter=300)
    assert np.array_equal(x, xr) and st.f_value == ref.f_value


def test_sann_custom_objective_equals_builtin(mod):
    rosen = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
    for n in (6, 200):
        x0 = starts(4, n, 0.5, 1.0)
        out = []
        for obj in ("rosenbrock", mod.CustomObjective(rosen, chain=True)):
            with mod.SANNEngine(obj, 4, n, max_iter=25, seed=5) as eng:
                x, st = eng.minimize(x0)
            out.append((x, [(s.f_value, s.function_calls_used) for s in st]))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_sann_rejects_bad_configs(mod):
    from nlsolver_amd._capi import NlsgError
    with pytest.raises(NlsgError):
        mod.SANNEngine("rosenbrock", 0, 4)
    with pytest.raises(NlsgError):
        mod.SANNEngine(17, 1, 4)
    with pytest.raises(NlsgError):  # a whole-vector body needs the point in registers
        mod.SANNEngine(mod.CustomObjective("return x(0) * x(0);", vector=True), 1, 2000)


@pytest.mark.parametrize("objective,n,kw", [
    ("rosenbrock", 1025, dict(max_iter=6, temperature_iter=5, temperature_max=10.0)),
    ("rosenbrock", 1026, dict(max_iter=6, temperature_iter=5, temperature_max=1e5)),
    ("sphere", 2048, dict(max_iter=5, temperature_iter=4, temperature_max=1e3)),
    ("styblinski_tang", 3001, dict(max_iter=4, temperature_iter=4, temperature_max=1e5)),
    ("rastrigin", 2049, dict(max_iter=4, temperature_iter=4, temperature_max=1e3)),
])
@pytest.mark.parametrize("minimize", [True, False])
def test_sann_long_chains_bit_exact(mod, oracle, objective, n, kw, minimize):
    """Chains past the register-resident layout (n > 1024; the reference has no limit,
    nlsolver.h:2777-2814): points streamed from memory in segments, same draws and same
    whole-row summation order. At the high temperatures worse trials are accepted too, so the
    current point leaves the best one (the histories differ from the temperature_max = 10 ones
    in every maximize case: checked against the oracle)."""
    batch = 5
    x0 = starts(batch, n, 0.5, 1.0)
    with mod.SANNEngine(objective, batch, n, minimize=minimize, seed=SEED, chain_lo=1, **kw) as eng:
        x, st = eng.minimize(x0)
    for b in range(batch):
        ref, xr, _ = O.sann_sync(oracle, objective, x0[b], SEED, 1 + b, minimize=minimize,
                                 max_iter=kw["max_iter"], temp_iter=kw["temperature_iter"],
                                 temp_max=kw["temperature_max"])
        assert np.array_equal(x[b], xr), (b, np.flatnonzero(x[b] != xr)[:8])
        assert st[b].f_value == ref.f_value
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)


def test_sann_long_chain_custom_objective_equals_builtin(mod):
    rosen = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
    n = 1500
    x0 = starts(3, n, 0.5, 1.0)
    out = []
    for obj in ("rosenbrock", mod.CustomObjective(rosen, chain=True)):
        with mod.SANNEngine(obj, 3, n, max_iter=5, temperature_iter=4, temperature_max=1e5,
                            seed=5) as eng:
            x, st = eng.minimize(x0)
        out.append((x, [(s.f_value, s.function_calls_used) for s in st]))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_sann_bench_size_properties(mod, oracle):
    """The bench configuration (Rosenbrock-128D, 16 384 chains, 40 temperatures here): runs are
    deterministic, a chain's best never exceeds its start, the reported value is the objective at
    the returned point, and sampled chains equal the oracle."""
    B, n = 16384, 128
    rng = np.random.default_rng(5)
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    kw = dict(max_iter=40, temperature_iter=10, temperature_max=10.0, seed=SEED)
    with mod.SANNEngine("rosenbrock", B, n, **kw) as eng:
        xa, sa = eng.minimize(x0)
        xb, sb = eng.minimize(x0)
    assert np.array_equal(xa, xb) and [s.f_value for s in sa] == [s.f_value for s in sb]
    f0 = ((1 - x0[:, :-1]) ** 2 + 100 * (x0[:, 1:] - x0[:, :-1] ** 2) ** 2).sum(axis=1)
    fa = np.array([s.f_value for s in sa])
    assert np.all(fa <= f0 * (1 + 1e-12))
    assert all(s.function_calls_used == 1 + 40 * 9 for s in sa)
    for b in (0, 4097, B - 1):
        ref, xr, _ = O.sann_sync(oracle, "rosenbrock", x0[b], SEED, b, max_iter=40, temp_iter=10,
                                 temp_max=10.0)
        assert np.array_equal(xa[b], xr) and sa[b].f_value == ref.f_value
        assert sa[b].f_value == oracle.orc_objective_tree(0, xa[b].ctypes.data_as(O.pd), n)
