"""The reference-order modes (NLSG_BFGS_REFERENCE_ORDER, NLSG_LM_CHOLESKY_REFERENCE_ORDER): every
sum in index order — the reference's sequential loops — instead of the kernels' lane tree, the
LM solve with the reference's separate multiply and add. With them the default-functor models
(fin_diff, fin_diff_h: nlsolver.h:1385-1517) reproduce the reference's OWN runs
(tests/golden/bfgs_fd.json, lm_fd.json, bfgs.json: outputs of the unmodified reference) BIT FOR
BIT — counts, iterates and objective values — where the tree order agrees to 1e-8 .. 1e-6 only
(finite differences divide the last bit of a sum by 12 eps or 600 eps^2)."""
import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import hx

pytestmark = pytest.mark.gpu
NAMES = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}


@pytest.fixture(scope="module")
def mod():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd


def start(g):
    return hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)


@pytest.mark.parametrize("name", ["rosenbrock_n2", "rosenbrock_n4", "rosenbrock_n16_default_stop",
                                  "rosenbrock_n128_20iters", "sphere_n5", "sphere_n130_alpha_half",
                                  "styblinski_tang_n8"])
def test_bfgs_default_gradient_equals_the_reference_run(mod, golden, name):
    """BFGS with the reference's default gradient on the device, reference order: the reference's
    run itself — every count, f and x bit for bit (north_star asks for 1e-12; this is 0)."""
    g = golden("bfgs_fd.json")[name]
    x = start(g).reshape(1, -1)
    with mod.BFGSEngine(NAMES[g["objective"]], 1, dim=g["n"], max_iter=g["max_iter"],
                        grad_eps=hx(g["grad_eps"]), alpha=hx(g["alpha"]), reference_order=True) as eng:
        x, st = eng.minimize(x)
    assert (st[0].iteration, st[0].function_calls_used, st[0].gradient_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"])
    assert st[0].f_value == hx(g["f"])
    assert x[0].tolist() == [hx(v) for v in g["x"]]


@pytest.mark.parametrize("name", ["n8", "n64", "n64_default_stop", "n100_ragged_start",
                                  "n130_alpha_half", "n256_max_iter_5", "n1024"])
def test_bfgs_analytic_gradient_equals_the_reference_run(mod, golden, name):
    """The G6 quadratic with its analytic gradient functor (tests/golden/bfgs.json), reference
    order: counts and the final value of the reference's run bit for bit — n = 1024, the dimension of
    BASELINE configs[2], included (37 iterations, 158 function and gradient calls)."""
    g = golden("bfgs.json")[name]
    n = g["n"]
    d, b, c = O.quad_problem(n)
    x = start(g).reshape(1, -1)
    with mod.BFGSEngine(mod.QuadDiagRank1(d, b, c), 1, max_iter=g["max_iter"],
                        grad_eps=hx(g["grad_eps"]), alpha=hx(g["alpha"]), reference_order=True) as eng:
        x, st = eng.minimize(x)
    assert (st[0].iteration, st[0].function_calls_used, st[0].gradient_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"])
    assert st[0].f_value == hx(g["f"])
    assert x[0][:8].tolist() == [hx(v) for v in g["x_head"]]
    h = 1469598103934665603  # FNV-1a over the bytes of the whole x, as the reference driver hashed it
    for byte in np.ascontiguousarray(x[0]).view(np.uint8).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert h == int(g["x_fnv"])


@pytest.mark.parametrize("obj,n,batch", [("rosenbrock", 3, 5), ("rosenbrock", 31, 4), ("sphere", 129, 3),
                                         ("styblinski_tang", 200, 2), ("rosenbrock", 256, 2)])
def test_bfgs_reference_order_batches_equal_the_serial_oracle(mod, oracle, obj, n, batch):
    """Random starts, odd sizes, two chunks: device in reference order == oracle tree 0 (the
    restatement that is pinned to the reference's runs), bit for bit."""
    kw = dict(max_iter=6, grad_eps=0.0, alpha=1.0)
    rng = np.random.default_rng(300 + n)
    x0 = 0.8 + 0.4 * (rng.random((batch, n)) - 0.5)
    with mod.BFGSEngine(obj, batch, dim=n, reference_order=True, **kw) as eng:
        x, st = eng.minimize(x0.copy())
    for p in range(batch):
        ref, xr, _, _ = O.bfgs_fd(oracle, obj, x0[p], tree=0, **kw)
        assert (st[p].iteration, st[p].function_calls_used, st[p].gradient_evals_used) == \
            (ref.iteration, ref.function_calls_used, ref.gradient_evals_used), p
        assert st[p].f_value == ref.f_value and np.array_equal(x[p], xr), p


def test_bfgs_reference_order_limits(mod):
    for args, kw in ((("rastrigin", 1), dict(dim=4)),                          # no libm cosine on the device
                     (("rosenbrock", 1), dict(dim=4, symmetric=True))):        # a different arithmetic
        with pytest.raises(mod.NlsgError):
            mod.BFGSEngine(*args, reference_order=True, **kw)


def test_lm_default_functors_equal_the_reference_runs(mod, golden):
    """LevenbergMarquardt with its default functors (fin_diff + fin_diff_h) on the device in
    reference order: the seven committed runs of the reference bit for bit, the NaN run included."""
    from nlsolver_amd._capi import LM_CHOLESKY_REFERENCE_ORDER
    for name, g in golden("lm_fd.json").items():
        x = start(g)
        solver = mod.lm.LevenbergMarquardt(NAMES[g["objective"]], hx(g["lambda"]), 10.0, 10.0,
                                           g["max_iter"], hx(g["f_delta"]),
                                           solver=LM_CHOLESKY_REFERENCE_ORDER)
        st = solver.minimize(x)
        f_ref, x_ref = hx(g["f"]), np.array([hx(v) for v in g["x"]])
        assert (st.iteration, st.function_calls_used, st.gradient_evals_used, st.hessian_evals_used) == \
            (g["iters"], g["fcalls"], g["gcalls"], g["hcalls"]), name
        if np.isnan(f_ref):
            assert np.isnan(st.f_value) and np.isnan(x).all(), name
            continue
        assert st.f_value == f_ref, (name, st.f_value, f_ref)
        assert np.array_equal(x, x_ref), name


@pytest.mark.parametrize("obj,n,batch", [("rosenbrock", 3, 6), ("rosenbrock", 9, 4), ("sphere", 17, 3),
                                         ("styblinski_tang", 33, 2), ("rosenbrock", 64, 2),
                                         # past 64 parameters: the workgroup-per-problem kernels (round 4)
                                         ("rosenbrock", 65, 2), ("sphere", 129, 2),
                                         ("styblinski_tang", 200, 1), ("rosenbrock", 257, 1)])
def test_lm_reference_order_batches_equal_the_serial_oracle(mod, oracle, obj, n, batch):
    """Every group width (4, 8, 16, 32 lanes per probe point): device == oracle order 0."""
    from nlsolver_amd._capi import LM_CHOLESKY_REFERENCE_ORDER
    kw = dict(lam=10.0, max_iter=3 if n <= 64 else 2, f_delta=0.0)
    rng = np.random.default_rng(500 + n)
    x0 = 0.9 + 0.2 * (rng.random((batch, n)) - 0.5)
    with mod.lm.LMEngine(obj, batch=batch, n=n, solver=LM_CHOLESKY_REFERENCE_ORDER, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_fd(oracle, obj, x0[b], order=0, **kw)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)
        assert st[b].f_value == ref.f_value and np.array_equal(x[b], xr) and lam[b] == lam_r, b


def test_lm_reference_order_limits(mod):
    from nlsolver_amd._capi import LM_CHOLESKY_REFERENCE_ORDER, NlsgError
    with pytest.raises(NlsgError):  # (n <= 1024 like every LM engine; 70 is served since round 4)
        mod.lm.LMEngine("rosenbrock", batch=1, n=1100, solver=LM_CHOLESKY_REFERENCE_ORDER)
    with pytest.raises(NlsgError):
        mod.lm.LMEngine("rastrigin", batch=1, n=4, solver=LM_CHOLESKY_REFERENCE_ORDER)
    with pytest.raises(NlsgError):  # a whole-vector body's x.sum() adds in the lane-tree order
        mod.lm.LMEngine(mod.CustomObjective("return x(0) * x(0) + x(1) * x(1);", vector=True), batch=1, n=2,
                        solver=LM_CHOLESKY_REFERENCE_ORDER)


def test_drop_in_classes_default_to_reference_order(mod, oracle, golden):
    """The drop-in classes' default (include/nlsolver_mi/nlsolver.h device::summation()): reference order
    wherever the reference's arithmetic exists on the device — one start or a batch returns the
    reference's run bit for bit; reference_order=False / solver=LM_CHOLESKY select the tree kernels."""
    g = golden("bfgs_fd.json")["rosenbrock_n16_default_stop"]
    kw = dict(max_iter=g["max_iter"], grad_eps=hx(g["grad_eps"]), alpha=hx(g["alpha"]))
    x = start(g)
    st = mod.BFGS("rosenbrock", None, **kw).minimize(x)
    assert st.f_value == hx(g["f"]) and x.tolist() == [hx(v) for v in g["x"]]
    assert (st.iteration, st.function_calls_used) == (g["iters"], g["fcalls"])
    xb = start(g).reshape(1, -1)  # (with the default gradient a batch solves in reference order too)
    stb = mod.BFGS("rosenbrock", None, **kw).minimize(xb)
    assert stb[0].f_value == st.f_value and np.array_equal(xb[0], x)
    xb = start(g).reshape(1, -1)
    stb = mod.BFGS("rosenbrock", None, reference_order=False, **kw).minimize(xb)
    tree, xt, _, _ = O.bfgs_fd(oracle, "rosenbrock", start(g), tree=1, **kw)
    assert stb[0].f_value == tree.f_value != st.f_value and np.array_equal(xb[0], xt)
    # the quadratic with its gradient functor: reference order too, one start or a batch
    gq = golden("bfgs.json")["n64"]
    d, b, c = O.quad_problem(gq["n"])
    kq = dict(max_iter=gq["max_iter"], grad_eps=hx(gq["grad_eps"]), alpha=hx(gq["alpha"]))
    xq = start(gq)
    sq = mod.BFGS(mod.QuadDiagRank1(d, b, c), None, **kq).minimize(xq)
    assert sq.f_value == hx(gq["f"]) and xq[:8].tolist() == [hx(v) for v in gq["x_head"]]
    xqb = start(gq).reshape(1, -1)
    sqb = mod.BFGS(mod.QuadDiagRank1(d, b, c), None, **kq).minimize(xqb)
    assert sqb[0].f_value == sq.f_value and np.array_equal(xqb[0], xq)
    xqb = start(gq).reshape(1, -1)
    sqb = mod.BFGS(mod.QuadDiagRank1(d, b, c), None, reference_order=False, **kq).minimize(xqb)
    qt, xqt, _ = O.bfgs_quad(oracle, start(gq), tree=1, **kq)
    assert sqb[0].f_value == qt.f_value and np.array_equal(xqb[0], xqt)

    g = golden("lm_fd.json")["rosenbrock_n16_6iters"]
    args = (hx(g["lambda"]), 10.0, 10.0, g["max_iter"], hx(g["f_delta"]))
    x = start(g)
    st = mod.lm.LevenbergMarquardt("rosenbrock", *args).minimize(x)
    assert st.f_value == hx(g["f"]) and x.tolist() == [hx(v) for v in g["x"]]
    xb = start(g).reshape(1, -1)  # (n <= 64: a batch solves in reference order too)
    stb = mod.lm.LevenbergMarquardt("rosenbrock", *args).minimize(xb)
    assert stb[0].f_value == st.f_value and np.array_equal(xb[0], x)
    from nlsolver_amd._capi import LM_CHOLESKY
    xb = start(g).reshape(1, -1)
    stb = mod.lm.LevenbergMarquardt("rosenbrock", *args, solver=LM_CHOLESKY).minimize(xb)
    tree, xt, _, _ = O.lm_fd(oracle, "rosenbrock", start(g), lam=hx(g["lambda"]), max_iter=g["max_iter"],
                             f_delta=hx(g["f_delta"]), order=1)
    assert stb[0].f_value == tree.f_value != st.f_value and np.array_equal(xb[0], xt)
    # Rastrigin has no reference arithmetic on the device: one start still runs (tree order)
    x = np.full(4, 0.3)
    assert np.isfinite(mod.BFGS("rastrigin").minimize(x).f_value)


# ---- Nelder-Mead in reference order (NLSG_NM_REFERENCE_ORDER) -----------------------------------------
def _nm_case(g):
    D = g["D"]
    x0 = hx(g["x0"]) + hx(g["x0_step"]) * np.arange(D, dtype=np.float64)
    kw = dict(step=hx(g["step"]), eps=hx(g["eps"]), max_iter=g["max_iter"],
              no_change_best_tol=g["no_change"], restarts=g["restarts"], minimize=bool(g["minimize"]))
    bounds = (hx(g["upper"]), hx(g["lower"])) if g["bounded"] else ()
    return D, x0, kw, bounds


@pytest.mark.parametrize("name", ["example_2d", "d4_200iters", "d4_fixed_step", "d16_bounded", "d8_restarts",
                                  "d6_maximize_bounded", "d128_2000iters", "d130_ragged"])
def test_nm_reference_order_equals_the_reference_runs(mod, golden, name):
    """Every committed run of the reference's NelderMead on the device in reference order: counts, the
    final value and the best vertex BIT FOR BIT — the 128-D run of 203 457 evaluations (which the
    lane-tree sums fork from at its 262nd: a tie broken the other way) and the 130-D run whose collapsed
    simplex shrinks a summation-order-dependent number of times included."""
    g = golden("nm.json")[name]
    D, x0, kw, bounds = _nm_case(g)
    with mod.NMEngine("rosenbrock", 1, D, bounded=bool(bounds), reference_order=True, **kw) as eng:
        x, st, _ = eng.minimize(x0[None].copy(), *bounds)
    assert (st[0].function_calls_used, st[0].iteration) == (g["fcalls"], g["iters"])
    assert st[0].f_value == hx(g["f"])
    assert x[0].tolist() == [hx(v) for v in g["x"]]


@pytest.mark.parametrize("obj,n,batch", [("rosenbrock", 3, 5), ("sphere", 17, 4), ("styblinski_tang", 64, 3),
                                         ("rosenbrock", 100, 2), ("rosenbrock", 200, 2), ("sphere", 300, 1)])
def test_nm_reference_order_batches_equal_the_serial_oracle(mod, oracle, obj, n, batch):
    """Random starts, every simplex size class (LDS-resident and the global workspace past 128):
    device in reference order == oracle order 0 (pinned to the reference's runs), bit for bit."""
    rng = np.random.default_rng(700 + n)
    x0 = 0.5 + 1.0 * (rng.random((batch, n)) - 0.5)
    kw = dict(step=-1.0, eps=0.0, max_iter=60, no_change_best_tol=10**6, restarts=0)
    with mod.NMEngine(obj, batch, n, reference_order=True, **kw) as eng:
        x, st, _ = eng.minimize(x0.copy())
    for b in range(batch):
        ref, xr, _, _ = O.nm_run(oracle, x0[b], obj=obj, order=0, step=-1.0, eps=0.0, max_iter=60,
                                 no_change=10**6, restarts=0)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used), b
        assert st[b].f_value == ref.f_value and np.array_equal(x[b], xr), b


def test_nm_reference_order_limits_and_default(mod, golden):
    with pytest.raises(mod.NlsgError):
        mod.NMEngine("rastrigin", 1, 4, reference_order=True)
    with pytest.raises(mod.NlsgError):
        mod.NMEngine(mod.CustomObjective("return x(0) * x(0) + x(1) * x(1);", vector=True), 1, 2,
                     reference_order=True)
    # the drop-in class: reference order by default — the reference's example run bit for bit
    g = golden("nm.json")["example_2d"]
    x = np.array([2.0, 7.0])
    st = mod.NelderMead("rosenbrock").minimize(x)
    assert (st.function_calls_used, st.iteration) == (g["fcalls"], g["iters"])
    assert st.f_value == hx(g["f"]) and x.tolist() == [hx(v) for v in g["x"]]
    # the same chain as source text
    x = np.array([2.0, 7.0])
    obj = mod.CustomObjective("double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;",
                              chain=True)
    st = mod.NelderMead(obj).minimize(x)
    assert st.f_value == hx(g["f"]) and x.tolist() == [hx(v) for v in g["x"]]


# ---- edges: one and two coordinates, the largest sizes ------------------------------------------------
@pytest.mark.parametrize("obj,n", [("sphere", 1), ("rosenbrock", 1), ("rosenbrock", 2), ("styblinski_tang", 1),
                                   ("rosenbrock", 1000), ("sphere", 1024)])
def test_bfgs_reference_order_edge_sizes(mod, oracle, obj, n):
    kw = dict(max_iter=4 if n > 64 else 12, grad_eps=0.0, alpha=1.0)
    rng = np.random.default_rng(900 + n)
    x0 = 0.8 + 0.4 * (rng.random((2, n)) - 0.5)
    with mod.BFGSEngine(obj, 2, dim=n, reference_order=True, **kw) as eng:
        x, st = eng.minimize(x0.copy())
    for p in range(2):
        ref, xr, _, _ = O.bfgs_fd(oracle, obj, x0[p], tree=0, **kw)
        assert (st[p].iteration, st[p].function_calls_used, st[p].gradient_evals_used) == \
            (ref.iteration, ref.function_calls_used, ref.gradient_evals_used), p
        same_f = st[p].f_value == ref.f_value or (np.isnan(st[p].f_value) and np.isnan(ref.f_value))
        assert same_f and np.array_equal(x[p], xr, equal_nan=True), p


@pytest.mark.parametrize("obj,n", [("sphere", 1), ("rosenbrock", 1), ("rosenbrock", 2), ("styblinski_tang", 2)])
def test_lm_and_nm_reference_order_edge_sizes(mod, oracle, obj, n):
    from nlsolver_amd._capi import LM_CHOLESKY_REFERENCE_ORDER
    rng = np.random.default_rng(950 + n)
    x0 = 0.9 + 0.2 * (rng.random((2, n)) - 0.5)
    kw = dict(lam=10.0, max_iter=4, f_delta=0.0)
    with mod.lm.LMEngine(obj, batch=2, n=n, solver=LM_CHOLESKY_REFERENCE_ORDER, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for b in range(2):
        ref, xr, lam_r, _ = O.lm_fd(oracle, obj, x0[b], order=0, **kw)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)
        same_f = st[b].f_value == ref.f_value or (np.isnan(st[b].f_value) and np.isnan(ref.f_value))
        assert same_f and np.array_equal(x[b], xr, equal_nan=True), b
    with mod.NMEngine(obj, 2, n, reference_order=True, step=-1.0, eps=0.0, max_iter=30,
                      no_change_best_tol=10**6) as eng:
        x, st, _ = eng.minimize(x0.copy())
    for b in range(2):
        ref, xr, _, _ = O.nm_run(oracle, x0[b], obj=obj, order=0, step=-1.0, eps=0.0, max_iter=30,
                                 no_change=10**6, restarts=0)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used), b
        same_f = st[b].f_value == ref.f_value or (np.isnan(st[b].f_value) and np.isnan(ref.f_value))
        assert same_f and np.array_equal(x[b], xr, equal_nan=True), b
