"""User objectives compiled at engine creation (nlsg_de_create_custom).

The run-time compiled kernels are the same templates as the built-in ones, instantiated around a
user-written term/finish pair, so a custom objective that spells a built-in one must reproduce
that engine (and hence the oracle) bit for bit; an objective nobody built in is checked against a
numpy evaluation of the same sum."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu

ROSENBROCK = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
STYBLINSKI = "double x2 = xi * xi; return x2 * x2 - 16 * x2 + 5 * xi;"


@pytest.fixture(scope="module")
def mod():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd


def run(mod, objective, pop, D, x0, turns, **kw):
    with mod.DEEngine(objective, pop, D, **kw) as eng:
        eng.init(x0)
        eng.step(turns)
        P, S = eng.download()
        st = eng.status()
        bx, bf, bi = eng.best()
    return P, S, st, bx, bf, bi


@pytest.mark.parametrize("D,pop", [(2, 64), (16, 256), (128, 2048), (130, 512), (257, 128)])
@pytest.mark.parametrize("strategy", [0, 1])
def test_custom_rosenbrock_equals_builtin_bit_for_bit(mod, oracle, D, pop, strategy):
    kw = dict(strategy=strategy, CR=0.2, F=0.5, eps=1e-300, max_iter=1000, best_val_no_change=1000,
              seed=4242)
    x0 = np.full(D, 0.6)
    a = run(mod, "rosenbrock", pop, D, x0, 12, **kw)
    b = run(mod, mod.CustomObjective(ROSENBROCK, chain=True), pop, D, x0, 12, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[2].iteration, a[2].function_calls_used, a[2].best_index, a[2].std_err) == \
        (b[2].iteration, b[2].function_calls_used, b[2].best_index, b[2].std_err)
    assert np.array_equal(a[3], b[3]) and a[4] == b[4]
    # and therefore the oracle
    ref = O.DESyncRun(oracle, "rosenbrock", pop, D, x0, **kw)
    ref.step(12)
    assert np.array_equal(b[0], ref.population) and np.array_equal(b[1], ref.scores)


def test_custom_objective_on_long_rows(mod):
    """D = 1500 (> 1024: segment-streaming kernels, compiled at run time as well)."""
    kw = dict(strategy=1, CR=0.2, F=0.5, eps=0.0, max_iter=1000, best_val_no_change=1000, seed=5)
    D, pop = 1500, 48
    x0 = np.full(D, 0.6)
    a = run(mod, "rosenbrock", pop, D, x0, 5, **kw)
    b = run(mod, mod.CustomObjective(ROSENBROCK, chain=True), pop, D, x0, 5, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]


def test_custom_finish_and_maximize(mod):
    """Styblinski-Tang spells its halving in `finish`; maximize flips the sign like the built-in."""
    kw = dict(minimize=False, CR=0.5, F=0.7, eps=0.0, max_iter=1000, best_val_no_change=1000, seed=9)
    D, pop = 64, 512
    x0 = np.full(D, 3.0)
    a = run(mod, "styblinski_tang", pop, D, x0, 8, **kw)
    b = run(mod, mod.CustomObjective(STYBLINSKI, finish_body="return s / 2.0;"), pop, D, x0, 8, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[5] == b[5]


def test_objective_nobody_built_in(mod):
    """f(x) = sqrt(sum_i (|x_i|^3 + 0.1 x_i x_{i+1})): every score equals the numpy value of the
    same expression within rounding of the different summation order, and DE decreases it."""
    D, pop = 96, 1024
    obj = mod.CustomObjective("return fabs(xi) * xi * xi + 0.1 * xi * xn;", chain=True,
                              finish_body="return sqrt(s > 0 ? s : 0.0);")
    x0 = np.full(D, 2.0)
    with mod.DEEngine(obj, pop, D, CR=0.3, F=0.5, eps=0.0, max_iter=10**6,
                      best_val_no_change=10**6, seed=3) as eng:
        eng.init(x0)
        P0, S0 = eng.download()
        eng.step(60)
        P, S = eng.download()
        bx, bf, bi = eng.best()
    def f(X):
        s = np.sum(np.abs(X[:, :-1]) ** 3 + 0.1 * X[:, :-1] * X[:, 1:], axis=1)
        return np.sqrt(np.maximum(s, 0.0))
    assert np.allclose(S0, f(P0), rtol=1e-13, atol=0) and np.allclose(S, f(P), rtol=1e-13, atol=0)
    assert bf == S[bi] and S.min() < S0.min() and np.mean(S < S0) > 0.9


def test_source_that_does_not_compile_is_reported(mod):
    with pytest.raises(RuntimeError, match="does not compile"):
        mod.DEEngine(mod.CustomObjective("return xi +;"), 64, 8)


@pytest.mark.parametrize("ptype", ["accelerated", "vanilla"])
@pytest.mark.parametrize("D,n", [(16, 256), (256, 1024), (130, 96)])
def test_pso_custom_rosenbrock_equals_builtin_bit_for_bit(mod, ptype, D, n):
    """The same hook in the PSO engine (nlsg_pso_create_custom): move + evaluate kernels compiled
    around the user's objective reproduce the built-in engine exactly."""
    kw = dict(type=mod.PSO_ACCELERATED if ptype == "accelerated" else mod.PSO_VANILLA, bounded=True,
              inertia=0.8, cognitive=1.8, social=1.8, eps=0.0, max_iter=1000,
              best_val_no_change=1000, seed=11)
    out = []
    for obj in ("rosenbrock", mod.CustomObjective(ROSENBROCK, chain=True)):
        with mod.PSOEngine(obj, n, D, **kw) as eng:
            eng.init(-2.048, 2.048)
            eng.step(10)
            st = eng.status()
            bx, bf, bi = eng.best()
            dl = eng.download()
        out.append((st.iteration, st.function_calls_used, bi, bf, bx, dl))
    a, b = out
    assert a[:4] == b[:4] and np.array_equal(a[4], b[4])
    for u, v in zip(a[5], b[5]):
        assert u is None and v is None or np.array_equal(u, v)


@pytest.mark.parametrize("n", [4, 16, 64, 128])
def test_nm_custom_rosenbrock_equals_builtin_bit_for_bit(mod, n):
    """Nelder-Mead with a run-time compiled objective (nlsg_nm_create_custom), up to the largest
    simplex (n = 128: 132 KiB of dynamic LDS through the module launch)."""
    rng = np.random.default_rng(n)
    x0 = 0.5 + 0.2 * (rng.random((6, n)) - 0.5)
    out = []
    for obj in ("rosenbrock", mod.CustomObjective(ROSENBROCK, chain=True)):
        with mod.NMEngine(obj, 6, n, eps=0.0, max_iter=120, no_change_best_tol=10**9) as eng:
            x, st, eps = eng.minimize(x0.copy())
        out.append((x, [(s.f_value, s.iteration, s.function_calls_used) for s in st]))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


@pytest.mark.parametrize("n", [4, 128, 200])
def test_bfgs_finite_difference_on_custom_objective_equals_builtin(mod, oracle, n):
    """BFGS with the default finite-difference gradient around a run-time compiled objective
    (nlsg_bfgs_create_custom): same iterates, value and counters as the built-in model, hence
    as the tree oracle."""
    rng = np.random.default_rng(n)
    x0 = 0.8 + 0.4 * (rng.random((3, n)) - 0.5)
    kw = dict(max_iter=5, grad_eps=0.0, alpha=1.0)
    with mod.BFGSEngine("rosenbrock", 3, dim=n, **kw) as eng:
        xa, sa = eng.minimize(x0.copy())
    with mod.BFGSEngine(mod.CustomObjective(ROSENBROCK, chain=True), 3, dim=n, **kw) as eng:
        xb, sb = eng.minimize(x0.copy())
    assert np.array_equal(xa, xb)
    for a, b in zip(sa, sb):
        assert (a.f_value, a.iteration, a.function_calls_used, a.gradient_evals_used) == \
            (b.f_value, b.iteration, b.function_calls_used, b.gradient_evals_used)
    ref, xr, _, _ = O.bfgs_fd(oracle, "rosenbrock", x0[0], tree=1, **kw)
    assert np.array_equal(xb[0], xr) and sb[0].f_value == ref.f_value


@pytest.mark.parametrize("n", [2, 7, 16])
def test_lm_default_functors_on_custom_objective_equals_builtin(mod, oracle, n):
    """LevenbergMarquardt with the default fin_diff / fin_diff_h functors around a run-time
    compiled objective (nlsg_lm_create_custom): same iterates as the built-in, hence the oracle."""
    rng = np.random.default_rng(n)
    x0 = 0.8 + 0.4 * (rng.random((4, n)) - 0.5)
    kw = dict(lam=10.0, max_iter=6, f_delta=0.0)
    with mod.lm.LMEngine("rosenbrock", batch=4, n=n, **kw) as eng:
        xa, sa, la = eng.minimize(x0.copy())
    with mod.lm.LMEngine(mod.CustomObjective(ROSENBROCK, chain=True), batch=4, n=n, **kw) as eng:
        xb, sb, lb = eng.minimize(x0.copy())
    assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(la, lb, equal_nan=True)
    for a, b in zip(sa, sb):
        assert (a.iteration, a.function_calls_used) == (b.iteration, b.function_calls_used)
        assert np.array_equal(a.f_value, b.f_value, equal_nan=True)
    ref, xr, lam_r, _ = O.lm_fd(oracle, "rosenbrock", x0[0], order=1, **kw)
    assert np.array_equal(xb[0], xr, equal_nan=True) and lb[0] == lam_r


def test_lm_custom_objective_quartic(mod):
    """An objective the library does not ship: f = sum (x_i^2 - 1)^2 + 0.1 x_i, minimised by the
    damped Newton iteration from its finite-difference model."""
    obj = mod.CustomObjective("double t = xi * xi - 1; return t * t + 0.1 * xi;")
    x0 = np.full((2, 5), 1.3)
    x0[1] = -0.7
    solver = mod.lm.LevenbergMarquardt(obj, 1.0, 10.0, 10.0, 60, 1e-14)
    st = solver.minimize(x0)
    # stationary points of t^2 + 0.1 x: 4 x (x^2 - 1) + 0.1 = 0
    r = 4 * x0 * (x0 * x0 - 1) + 0.1
    assert np.max(np.abs(r)) < 1e-4, (x0, st[0].f_value)


# ---- whole-vector objectives (NLSG_CUSTOM_VECTOR): the reference's test set is mostly made of
# ---- functions that are not sums of per-coordinate terms (test_functions.h:52-330)
HIMMELBLAU = ("double a = x(0) * x(0) + x(1) - 11, b = x(0) + x(1) * x(1) - 7;"
              " return a * a + b * b;")  # test_functions.h:131-135
BOOTH = "double a = x(0) + 2 * x(1) - 7, b = 2 * x(0) + x(1) - 5; return a * a + b * b;"  # :291-294
MATYAS = "return 0.26 * (x(0) * x(0) + x(1) * x(1)) - 0.48 * x(0) * x(1);"  # :311-313
# the Rosenbrock chain spelled with the accessor and a loop over a wave-uniform index (slow but
# legal): the terms are added sequentially — the reference's own order
ROSENBROCK_LOOP = ("double s = 0; for (uint64_t i = 0; i + 1 < D; i++) { double t1 = 1 - x(i);"
                   " double t2 = x(i + 1) - x(i) * x(i); s += t1 * t1 + 100 * t2 * t2; } return s;")
# sum |x_i| * (i + 1) through x.sum: the index is available to the term
WEIGHTED = "return x.sum([](double xi, uint64_t i) { return fabs(xi) * (double)(i + 1); });"


def himmelblau(p):
    a = p[:, 0] * p[:, 0] + p[:, 1] - 11
    b = p[:, 0] + p[:, 1] * p[:, 1] - 7
    return a * a + b * b


@pytest.mark.parametrize("strategy", [0, 1])
def test_whole_vector_objective_scores_are_exact(mod, strategy):
    """A polynomial whole-vector objective has no summation tree at all: the device's scores equal
    the same expression evaluated by numpy bit for bit, at the initial population and after turns."""
    pop, D = 256, 2
    kw = dict(strategy=strategy, CR=0.7, F=0.6, eps=0.0, max_iter=1000, best_val_no_change=1000,
              seed=77)
    x0 = np.array([5.0, 7.0])
    P, S, st, bx, bf, bi = run(mod, mod.CustomObjective(HIMMELBLAU, vector=True), pop, D, x0, 0, **kw)
    assert np.array_equal(S, himmelblau(P))
    P, S, st, bx, bf, bi = run(mod, mod.CustomObjective(HIMMELBLAU, vector=True), pop, D, x0, 25, **kw)
    assert np.array_equal(S, himmelblau(P))
    assert bf == S.min() and np.array_equal(bx, P[bi]) and bf < 1.0  # and it does minimise


def test_whole_vector_loop_over_coordinates_matches_sequential_reference_arithmetic(mod, oracle):
    """x(i) with a loop counter, D = 130 (two chunks): the body adds the Rosenbrock terms
    sequentially — exactly the reference functor's arithmetic (orc_objective_seq), not the
    kernels' tree."""
    pop, D = 64, 130
    x0 = np.full(D, 0.9)
    P, S, *_ = run(mod, mod.CustomObjective(ROSENBROCK_LOOP, vector=True), pop, D, x0, 0, eps=0.0)
    seq = np.array([oracle.orc_objective_seq(0, np.ascontiguousarray(r).ctypes.data_as(O.pd), D)
                    for r in P])
    assert np.array_equal(S, seq)


@pytest.mark.parametrize("D", [3, 16, 64, 128, 200])
def test_whole_vector_sum_uses_the_lane_tree(mod, D):
    """x.sum(g) is the built-in objectives' reduction (per lane ascending, then the butterfly) and
    hands g the coordinate's index — packed agents (D <= 64: several per wave) and whole waves."""
    pop = 128
    x0 = np.full(D, 2.0)
    P, S, *_ = run(mod, mod.CustomObjective(WEIGHTED, vector=True), pop, D, x0, 0, eps=0.0)
    w = np.arange(1, D + 1, dtype=np.float64)
    ref = np.empty(pop)
    for a in range(pop):
        lanes = np.zeros(64)
        for e in range(D):
            lanes[(e % 128) // 2] += abs(P[a, e]) * w[e]
        off = 32
        while off >= 1:
            lanes = lanes + lanes[np.arange(64) ^ off]
            off //= 2
        ref[a] = lanes[0]
    assert np.array_equal(S, ref)


def test_whole_vector_objective_in_every_engine(mod):
    """The same body through PSO, Nelder-Mead, SANN, the hybrid, finite-difference BFGS and LM:
    each engine lands on a minimum of Himmelblau / Booth / Matyas (the reference's tolerance: 0.05)."""
    obj = mod.CustomObjective(HIMMELBLAU, vector=True)
    minima = np.array([[3.0, 2.0], [-2.805118, 3.131312], [-3.779310, -3.283186], [3.584428, -1.848126]])

    def near_a_minimum(x):
        return np.min(np.max(np.abs(minima - x), axis=1)) < 0.05

    x = np.array([-0.5, -0.5])
    st = mod.NelderMead(obj).minimize(x)
    assert near_a_minimum(x) and st.f_value < 1e-3
    x = np.array([[-0.5, -0.5]])
    st = mod.BFGS(obj).minimize(x)
    assert near_a_minimum(x[0])
    x = np.array([-0.5, -0.5])
    st = mod.lm.LevenbergMarquardt(obj).minimize(x)
    # the class is a damped Newton iteration with an always-accepted step (nlsolver.h:3465-3544):
    # from this start it lands on Himmelblau's local maximum, a stationary point all the same
    grad = np.array([4 * x[0] * (x[0] ** 2 + x[1] - 11) + 2 * (x[0] + x[1] ** 2 - 7),
                     2 * (x[0] ** 2 + x[1] - 11) + 4 * x[1] * (x[0] + x[1] ** 2 - 7)])
    assert np.max(np.abs(grad)) < 1e-4 and np.allclose(x, [-0.270845, -0.923039], atol=1e-5)
    x = np.array([4.0, 4.0])
    st = mod.PSO(obj, 7, n_particles=64, max_iter=500).minimize(x)
    assert near_a_minimum(x)
    x = np.array([1.0, 3.0])
    booth = mod.CustomObjective(BOOTH, vector=True)
    st = mod.NelderMeadPSO(booth, 11).minimize(x)
    assert np.max(np.abs(x - [1.0, 3.0])) < 0.05
    x = np.array([0.5, -0.5])
    st = mod.SANN(mod.CustomObjective(MATYAS, vector=True), 5).minimize(x)
    assert st.f_value <= 0.26 * 0.5 + 0.48 * 0.25  # never worse than the start


def test_whole_vector_compile_error_is_reported(mod):
    with pytest.raises(mod.NlsgError, match="does not compile"):
        mod.DEEngine(mod.CustomObjective("return x(0) +;", vector=True), 64, 2)
