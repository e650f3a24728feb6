"""Parity tests: batched HIP Nelder-Mead / PSO hybrid (one workgroup per instance) vs
oracle_nmpso.c's synchronous variant (keyed draws, objective and std_err trees): bit-exact best
particles, values and counters for every instance."""
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


def check(mod, oracle, objective, n, batch, x0, minimize=True, bounds=None, **kw):
    okw = dict(eps=kw.get("eps", 1e-6), max_iter=kw.get("max_iter", 1000),
               no_change=kw.get("no_change_best_iter", 20))
    for k in ("alpha", "gamma", "rho", "sigma", "inertia"):
        if k in kw:
            okw[k] = kw[k]
    if "cognitive" in kw:
        okw["cog"] = kw["cognitive"]
    if "social" in kw:
        okw["soc"] = kw["social"]
    lower, upper = bounds if bounds else (None, None)
    with mod.NMPSOEngine(objective, batch, n, minimize=minimize, bounded=bounds is not None,
                         seed=SEED, inst_lo=2, **kw) as eng:
        x, st = eng.minimize(x0, lower, upper)
    for b in range(batch):
        ref, xr, _ = O.nmpso_sync(oracle, objective, x0[b], SEED, 2 + b, minimize=minimize,
                                  upper=upper, lower=lower, **okw)
        assert np.array_equal(x[b], xr), (b, x[b], xr)
        assert st[b].f_value == ref.f_value, (b, st[b].f_value, ref.f_value)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)
    return st


@pytest.mark.parametrize("objective,n,kw", [
    ("rosenbrock", 2, dict(max_iter=1000, eps=1e-6, no_change_best_iter=20)),
    ("rosenbrock", 3, dict(max_iter=60, eps=0.0, no_change_best_iter=1000)),
    ("rosenbrock", 4, dict(max_iter=100, eps=1e-6, no_change_best_iter=20)),
    ("rosenbrock", 8, dict(max_iter=120, eps=0.0, no_change_best_iter=1000)),
    ("sphere", 6, dict(max_iter=150, eps=1e-9, no_change_best_iter=20)),
    ("styblinski_tang", 5, dict(max_iter=60, eps=0.0, no_change_best_iter=1000)),
    ("rosenbrock", 16, dict(max_iter=80, eps=1e-6, no_change_best_iter=20)),
    ("rosenbrock", 33, dict(max_iter=40, eps=0.0, no_change_best_iter=1000)),
    ("rosenbrock", 64, dict(max_iter=30, eps=0.0, no_change_best_iter=1000)),
    ("rosenbrock", 127, dict(max_iter=12, eps=0.0, no_change_best_iter=1000)),
    ("sphere", 128, dict(max_iter=12, eps=0.0, no_change_best_iter=1000)),
])
def test_hybrid_instances_bit_exact_vs_sync_oracle(mod, oracle, objective, n, kw):
    check(mod, oracle, objective, n, 5, starts(5, n, 0.5, 1.0), **kw)


@pytest.mark.parametrize("objective,n,kw,bounded", [
    ("rosenbrock", 129, dict(max_iter=10, eps=0.0, no_change_best_iter=1000), False),
    ("sphere", 200, dict(max_iter=8, eps=0.0, no_change_best_iter=1000), True),
    ("styblinski_tang", 257, dict(max_iter=6, eps=0.0, no_change_best_iter=1000), False),
    ("rastrigin", 300, dict(max_iter=5, eps=1e-9, no_change_best_iter=3), False),
    ("rosenbrock", 513, dict(max_iter=4, eps=0.0, no_change_best_iter=1000), False),
    ("sphere", 1024, dict(max_iter=3, eps=0.0, no_change_best_iter=1000), True),
])
def test_hybrid_past_128_coordinates_bit_exact(mod, oracle, objective, n, kw, bounded):
    """n > 128 (the reference has no limit, nlsolver.h:3546-3920): the same kernel body over
    dynamic shared memory, one particle / pair per wave pass with the point in registers."""
    batch = 2
    x0 = starts(batch, n, 0.5, 1.0)
    bounds = (np.full(n, -1.5), np.full(n, 2.5)) if bounded else None
    check(mod, oracle, objective, n, batch, x0, bounds=bounds, **kw)


def test_hybrid_past_128_maximize_shrinks_and_custom(mod, oracle):
    """The shrink branch (sigma path, rescoring of n particles) and a run-time compiled objective
    at n = 200."""
    n = 200
    x0 = starts(2, n, 0.5, 1.0)
    check(mod, oracle, "sphere", n, 2, x0, minimize=False, max_iter=6, eps=0.0,
          no_change_best_iter=1000, alpha=0.4, gamma=1.1, rho=0.9, sigma=0.3)
    rosen = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
    for nn, iters in ((200, 5), (600, 2)):  # 600: 72 KiB of dynamic shared memory in the module kernel
        xs = starts(2, nn, 0.5, 1.0)
        out = []
        for obj in ("rosenbrock", mod.CustomObjective(rosen, chain=True)):
            with mod.NMPSOEngine(obj, 2, nn, max_iter=iters, eps=0.0, seed=5) as eng:
                x, st = eng.minimize(xs)
            out.append((x, [(s.f_value, s.function_calls_used) for s in st]))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    from nlsolver_amd._capi import NlsgError
    with pytest.raises(NlsgError):
        mod.NMPSOEngine("sphere", 1, 1025)


def test_hybrid_maximize_and_other_coefficients(mod, oracle):
    x0 = starts(4, 6, 0.5, 1.0)
    check(mod, oracle, "styblinski_tang", 6, 4, x0, minimize=False, max_iter=50, eps=0.0,
          no_change_best_iter=1000)
    check(mod, oracle, "rosenbrock", 6, 4, x0, max_iter=70, eps=0.0, no_change_best_iter=1000,
          alpha=1.2, gamma=1.8, rho=0.4, sigma=0.6, inertia=0.7, cognitive=1.5, social=1.2)


def test_hybrid_bounded_overload(mod, oracle):
    """minimize(x, lower, upper): simplex points and velocities clamped per coordinate."""
    x0 = starts(4, 8, 0.2, 0.8)
    st = check(mod, oracle, "styblinski_tang", 8, 4, x0, bounds=(-1.0, 1.0), max_iter=60, eps=0.0,
               no_change_best_iter=1000)
    assert all(np.isfinite(s.f_value) for s in st)


def test_hybrid_shrink_path_is_exercised(mod, oracle):
    """Small simplexes fail their contraction often: the shrink + rescore + re-sort path runs (the
    oracle counts its shrink steps; the device run agrees with it bit for bit)."""
    for objective, n in (("styblinski_tang", 4), ("rosenbrock", 2)):
        x0 = starts(4, n, -1.0, 3.0, seed=11)
        check(mod, oracle, objective, n, 4, x0, max_iter=300, eps=0.0, no_change_best_iter=10**6)
        assert oracle.orc_nmpso_last_shrinks() > 10  # of the last instance's oracle run


def test_hybrid_instance_ids_are_global(mod):
    x0 = starts(5, 6, 0.5, 1.0)
    kw = dict(max_iter=40, eps=0.0, seed=7)
    with mod.NMPSOEngine("rosenbrock", 5, 6, **kw) as eng:
        xa, sa = eng.minimize(x0)
    with mod.NMPSOEngine("rosenbrock", 3, 6, inst_lo=2, **kw) as eng:
        xb, sb = eng.minimize(x0[2:])
    assert np.array_equal(xa[2:], xb)
    assert [s.f_value for s in sa[2:]] == [s.f_value for s in sb]


def test_hybrid_converges_on_the_reference_example(mod):
    """The reference's own pass criterion (within 0.05 of the known minimum on a small problem)."""
    x = np.array([2.0, 7.0])
    st = mod.NelderMeadPSO("rosenbrock").minimize(x)
    assert np.all(np.abs(x - 1.0) <= 0.05), (x, st.f_value)


def test_hybrid_one_dimension_is_refused_like_the_reference(mod):
    from nlsolver_amd._capi import NlsgError
    st = mod.NelderMeadPSO("sphere").minimize(np.array([2.0]))
    assert (st.f_value, st.iteration, st.function_calls_used) == (999999, 0, 0)
    with pytest.raises(NlsgError):
        mod.NMPSOEngine("sphere", 1, 1)
    with pytest.raises(NlsgError):
        mod.NMPSOEngine("sphere", 1, 1025)


def test_hybrid_custom_objective_equals_builtin(mod):
    rosen = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
    for n in (4, 40):
        x0 = starts(3, n, 0.5, 1.0)
        out = []
        for obj in ("rosenbrock", mod.CustomObjective(rosen, chain=True)):
            with mod.NMPSOEngine(obj, 3, n, max_iter=30, eps=0.0, seed=5) as eng:
                x, st = eng.minimize(x0)
            out.append((x, [(s.f_value, s.function_calls_used) for s in st]))
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_hybrid_bench_size_properties(mod, oracle):
    """The bench configuration (Rosenbrock-32D, 4096 instances, 30 iterations here): deterministic,
    never worse than the best initial particle's value bound, value = objective at the returned
    point, sampled instances equal the oracle."""
    B, n = 4096, 32
    rng = np.random.default_rng(6)
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    kw = dict(max_iter=30, eps=0.0, no_change_best_iter=2**62, seed=SEED)
    with mod.NMPSOEngine("rosenbrock", B, n, **kw) as eng:
        xa, sa = eng.minimize(x0)
        xb, sb = eng.minimize(x0)
    assert np.array_equal(xa, xb) and [s.f_value for s in sa] == [s.f_value for s in sb]
    assert all(s.iteration == 30 for s in sa)
    for b in (0, 2049, B - 1):
        ref, xr, _ = O.nmpso_sync(oracle, "rosenbrock", x0[b], SEED, b, eps=0.0, max_iter=30,
                                  no_change=2**62)
        assert np.array_equal(xa[b], xr) and sa[b].f_value == ref.f_value
        assert sa[b].function_calls_used == ref.function_calls_used
        assert sa[b].f_value == oracle.orc_objective_tree(0, xa[b].ctypes.data_as(O.pd), n)


def test_a_later_smaller_engine_does_not_lower_the_lds_opt_in(mod, oracle):
    """n = 600, then n = 520 — the same kernel instantiation (eight chunks), whose dynamic-LDS
    opt-in is per instantiation, not per engine — then the first engine's launch."""
    kw = dict(max_iter=3, eps=0.0, no_change_best_iter=1000)
    x_big, x_small = starts(1, 600, 0.5, 1.0), starts(1, 520, 0.5, 1.0)
    big = mod.NMPSOEngine("rosenbrock", 1, 600, seed=SEED, inst_lo=2, **kw)
    small = mod.NMPSOEngine("rosenbrock", 1, 520, seed=SEED, inst_lo=2, **kw)
    xs, sts = small.minimize(x_small, None, None)
    xb, stb = big.minimize(x_big, None, None)
    small.close()
    big.close()
    for x0, x, st in ((x_big, xb, stb), (x_small, xs, sts)):
        ref, xr, _ = O.nmpso_sync(oracle, "rosenbrock", x0[0], SEED, 2, eps=0.0, max_iter=3,
                                  no_change=1000)
        assert np.array_equal(x[0], xr) and st[0].f_value == ref.f_value
