"""Parity tests: batched HIP Levenberg-Marquardt (J^T J on fp64 MFMA + Cholesky in LDS) vs
oracle_lm.c in the kernel's operation order (order=1): bit-exact parameters, objective
values, damping, counters. Against the reference run (golden, sequential sums, libm tanh):
objective within tolerance while above the rounding floor."""
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


def problems(oracle, first, batch, m, n):
    A, y, t0 = np.zeros((batch, m, n)), np.zeros((batch, m)), np.zeros((batch, n))
    for b in range(batch):
        A[b], y[b], t0[b] = O.tanh_problem(oracle, SEED, first + b, m, n)
    return A, y, t0


@pytest.mark.parametrize("m,n,batch", [(16, 4, 5), (64, 8, 6), (100, 33, 4), (128, 64, 6),
                                       (512, 64, 8), (70, 1, 3)])
@pytest.mark.parametrize("kw", [dict(lam=10.0, max_iter=20, f_delta=0.0),
                                dict(lam=10.0, max_iter=100, f_delta=1e-12),
                                dict(lam=0.5, up=4.0, down=3.0, max_iter=6, f_delta=0.0)])
def test_lm_batch_bit_exact_vs_kernel_order_oracle(mod, oracle, m, n, batch, kw):
    A, y, t0 = problems(oracle, 0, batch, m, n)
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in range(batch):
        ref, xr, lam_r, flog = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        assert (st[b].iteration, st[b].function_calls_used, st[b].gradient_evals_used,
                st[b].hessian_evals_used) == (ref.iteration, ref.function_calls_used,
                                              ref.gradient_evals_used, ref.hessian_evals_used), b
        assert st[b].f_value == ref.f_value, f"problem {b}"
        assert np.array_equal(th[b], xr), f"problem {b}"
        assert lam[b] == lam_r


def test_lm_matches_reference_run_within_tolerance(mod, oracle, golden):
    """Device result vs the reference LM class itself (golden f_vals of its own run; the serial
    oracle reproduces them bit for bit, so its parameter vectors stand for the reference's).

    Iteration 1 isolates rounding from amplification: f within 1e-12 relative (measured 4e-16;
    iterations 2 and 3: 1e-15, 2e-14), parameters within 1e-12 (measured <= 2.3e-16 absolute at
    EVERY iteration: the iterates never drift apart). What grows is only the RELATIVE error of
    f = sum r^2 as f itself collapses quadratically towards 0: rounding noise of size ~eps in
    the residuals is a relative error ~eps / sqrt(f / f0) of f. Measured law on this run:
    |f_dev - f_ref| / f_ref * sqrt(f_ref / f0) <= 4e-16 for every iteration down to the rounding
    floor (f ~ 5e-30 = m eps^2); asserted with a factor 25."""
    g = golden("lm.json")["tanh_m512_n64"]
    A, y, t0 = O.tanh_problem(oracle, SEED, 0, 512, 64)
    ref = np.array([float.fromhex(v) for v in g["f_vals"]])
    for k in (1, 2, 3, 4, 5, 6, 8, 12, 20):
        x = t0.copy()
        st = mod.LevenbergMarquardt(mod.TanhRegression(A, y), 10.0, 10.0, 10.0, k, 0.0).minimize(x)
        assert st.iteration == k
        ser, x_ser, _, _ = O.lm_solve(oracle, A, y, t0, order=0, lam=10.0, max_iter=k, f_delta=0.0)
        assert ser.f_value == ref[k]  # the stand-in IS the reference run
        assert np.max(np.abs(x - x_ser)) <= 1e-12
        if k <= 3:
            assert abs(st.f_value - ref[k]) <= 1e-12 * ref[k], k
        if ref[k] > 1e-28:
            assert abs(st.f_value - ref[k]) <= 1e-14 * np.sqrt(ref[k] * ref[0]), k
        else:  # both sit on the rounding floor of the residuals
            assert st.f_value < 1e-28
    assert float.fromhex(g["f"]) < 1e-20


def test_lm_config4_shape_sample(mod, oracle):
    """BASELINE config 4's shape (m=512, n=64) with a reduced batch, 20 iterations."""
    batch = 32
    A, y, t0 = problems(oracle, 100, batch, 512, 64)
    kw = dict(lam=10.0, max_iter=20, f_delta=0.0)
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in range(0, batch, 5):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        assert st[b].f_value == ref.f_value and np.array_equal(th[b], xr)
    assert all(s.iteration == 20 and s.f_value < 1e-20 for s in st)


@pytest.mark.parametrize("solver_name", ["cholesky", "qr"])
def test_lm_config4_full_batch(mod, oracle, solver_name):
    """BASELINE configs[3] at its full size — m = 512, n = 64, batch = 8192 (2 GiB of design
    matrices), 20 iterations, both solvers: 18 sampled problems including the first and the LAST
    index bit for bit against the kernel-order oracle; every problem ran its 20 iterations,
    reached the rounding floor, and its final damping is the one 20 down-steps from 10 give."""
    from nlsolver_amd import _capi
    m, n, batch = 512, 64, 8192
    rng = np.random.default_rng(20240 + (solver_name == "qr"))
    A = (2 * rng.random((batch, m, n)) - 1) / np.sqrt(n)
    star = 2 * rng.random((batch, n)) - 1
    y = np.tanh(np.einsum("bmn,bn->bm", A, star))
    t0 = 0.5 * star + 0.1 * (2 * rng.random((batch, n)) - 1)
    kw = dict(lam=10.0, max_iter=20, f_delta=0.0)
    solver = _capi.LM_QR if solver_name == "qr" else _capi.LM_CHOLESKY
    with mod.LMEngine(mod.TanhRegression(A, y), solver=solver, **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    sample = sorted({0, 1, 63, 64, 255, 256, 1023, 4095, 4096, 8190, 8191,
                     *rng.integers(0, batch, 7).tolist()})
    assert len(sample) >= 16
    for b in sample:
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1,
                                       solver=1 if solver_name == "qr" else 0, **kw)
        assert st[b].f_value == ref.f_value and np.array_equal(th[b], xr), b
        assert lam[b] == lam_r and st[b].iteration == ref.iteration, b
    f0 = np.array([np.sum((y[b] - np.tanh(A[b] @ t0[b])) ** 2) for b in range(0, batch, 97)])
    f = np.array([s.f_value for s in st])
    assert all(s.iteration == 20 and s.function_calls_used == 21 and s.done == 1 for s in st)
    assert np.all(np.isfinite(th)) and np.all(f < 1e-20) and np.all(f[::97] < f0)
    assert np.max(np.abs(th - star)) < 1e-6  # every problem found its generating parameters


@pytest.mark.parametrize("m,n,batch", [(16, 4, 4), (64, 8, 5), (100, 33, 3), (128, 64, 4), (512, 64, 6),
                                       (40, 2, 3)])
@pytest.mark.parametrize("kw", [dict(lam=10.0, max_iter=12, f_delta=0.0),
                                dict(lam=10.0, max_iter=100, f_delta=1e-12)])
def test_lm_qr_solver_bit_exact_vs_kernel_order_oracle(mod, oracle, m, n, batch, kw):
    """solver = QR: tinyqr::lm on the damped matrix (the composition BASELINE config 4 names),
    Givens rotations executed as wavefronts; bit-exact vs oracle tinyqr in order 1."""
    from nlsolver_amd import _capi
    A, y, t0 = problems(oracle, 40, batch, m, n)
    with mod.LMEngine(mod.TanhRegression(A, y), solver=_capi.LM_QR, **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], solver=1, order=1, **kw)
        assert (st[b].iteration, st[b].function_calls_used) == (ref.iteration, ref.function_calls_used)
        assert st[b].f_value == ref.f_value, f"problem {b}"
        assert np.array_equal(th[b], xr), f"problem {b}"
        assert lam[b] == lam_r


def test_lm_qr_and_cholesky_agree_to_rounding(mod, oracle):
    from nlsolver_amd import _capi
    A, y, t0 = problems(oracle, 7, 4, 256, 64)
    kw = dict(lam=10.0, max_iter=6, f_delta=0.0)
    with mod.LMEngine(mod.TanhRegression(A, y), solver=_capi.LM_QR, **kw) as eng:
        tq, sq, _ = eng.minimize(t0.copy())
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        tc, sc, _ = eng.minimize(t0.copy())
    assert np.allclose(tq, tc, rtol=1e-8, atol=1e-10) and not np.array_equal(tq, tc)


# ---- default functors (fin_diff + fin_diff_h) on built-in objectives, SURVEY.md §8f N2
def fd_starts(oracle, objective, batch, n, scale):
    x0 = np.zeros((batch, n))
    for b in range(batch):
        k = oracle.orc_ctr_key(SEED, 1000 + b)
        u = np.array([oracle.orc_u01(oracle.orc_ctr_key(k, j)) for j in range(n)])
        x0[b] = scale * (2 * u - 1)
    return x0


@pytest.mark.parametrize("objective,n,scale", [("rosenbrock", 2, 2.0), ("rosenbrock", 4, 1.5),
                                               ("rosenbrock", 7, 1.2), ("rosenbrock", 16, 1.0),
                                               ("sphere", 1, 3.0), ("sphere", 5, 3.0),
                                               ("styblinski_tang", 8, 4.0),
                                               ("styblinski_tang", 33, 4.0),
                                               ("rastrigin", 2, 4.0), ("rastrigin", 12, 30.0)])
@pytest.mark.parametrize("kw", [dict(lam=10.0, max_iter=12, f_delta=1e-12),
                                dict(lam=1.0, up=4.0, down=3.0, max_iter=5, f_delta=0.0)])
def test_lm_default_functors_bit_exact_vs_oracle(mod, oracle, objective, n, scale, kw):
    """fin_diff / fin_diff_h probes through the wave's objective tree (oracle order 1): parameters,
    objective value, damping, iteration and probe counts agree exactly, NaN runs included."""
    batch = 6
    x0 = fd_starts(oracle, objective, batch, n, scale)
    with mod.lm.LMEngine(objective, batch=batch, n=n, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_fd(oracle, objective, x0[b], order=1, **kw)
        assert np.array_equal(x[b], xr, equal_nan=True), (b, x[b], xr)
        assert np.array_equal(st[b].f_value, ref.f_value, equal_nan=True)
        assert np.array_equal(lam[b], lam_r, equal_nan=True)
        assert st[b].iteration == ref.iteration
        assert st[b].function_calls_used == ref.function_calls_used
        assert st[b].gradient_evals_used == ref.gradient_evals_used
        assert st[b].hessian_evals_used == ref.hessian_evals_used


def test_lm_default_functors_n64(mod, oracle):
    x0 = fd_starts(oracle, "sphere", 3, 64, 2.0)
    kw = dict(lam=1.0, max_iter=3, f_delta=0.0)
    with mod.lm.LMEngine("sphere", batch=3, n=64, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for b in range(3):
        ref, xr, lam_r, _ = O.lm_fd(oracle, "sphere", x0[b], order=1, **kw)
        assert np.array_equal(x[b], xr) and st[b].f_value == ref.f_value and lam[b] == lam_r
        assert st[b].function_calls_used == ref.function_calls_used == 4 * (1 + 4 * 64 + 16 * 64 * 64)


def test_lm_default_functors_match_reference_runs(mod, oracle, golden):
    """The committed runs of the reference's LevenbergMarquardt with its default functors
    (tests/golden/lm_fd.json); probe and iteration counts agree exactly.

    Why these are not 1e-12 even after ONE iteration: fin_diff_h divides differences of objective
    values by 600 eps^2 = 9e-6 (eps = DBL_EPSILON^(1/4)), so the last-bit difference between the
    reference's sequential sum and the wave's tree sum (~1e-16 |f|) enters the Hessian multiplied
    by ~1e5 before any iteration has amplified anything. Measured after iteration 1 (device order
    vs the serial oracle, which reproduces the reference bit for bit): f within 4e-11 (Rosenbrock-16),
    1e-9 (Styblinski-Tang-8), 3e-9 (Sphere-64) relative, 0 for the 2-D and 4-D runs (at most two
    terms per lane: tree and sequence coincide); over the following iterations the difference
    stays at 1e-9 .. 2e-7 — it does not compound, the always-accepted step contracts it. Asserted:
    1e-8 after iteration 1, 1e-6 at the end (the golden's own print precision is finer)."""
    from nlsolver_amd._capi import LM_CHOLESKY  # (the tree-order kernels; one start defaults to reference order)
    from tests.test_oracle_golden import hx
    names = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}
    for name, g in golden("lm_fd.json").items():
        if g["n"] > 64:
            # the runs past 64 parameters (round 4) pin the REFERENCE-ORDER mode, bit for bit
            # (tests/test_reference_order_gpu.py); the tolerances below were measured for n <= 64
            continue
        x0 = hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)
        x = x0.copy()
        solver = mod.lm.LevenbergMarquardt(names[g["objective"]], hx(g["lambda"]), 10.0, 10.0,
                                           g["max_iter"], hx(g["f_delta"]), solver=LM_CHOLESKY)
        st = solver.minimize(x)
        f_ref, x_ref = hx(g["f"]), np.array([hx(v) for v in g["x"]])
        if np.isnan(f_ref):
            assert np.isnan(st.f_value), name
            continue
        assert st.iteration == g["iters"], name
        assert st.function_calls_used == g["fcalls"], name
        assert st.gradient_evals_used == g["gcalls"] and st.hessian_evals_used == g["hcalls"], name
        assert np.allclose(x, x_ref, rtol=1e-5, atol=1e-6), name
        assert abs(st.f_value - f_ref) <= 1e-6 * max(1.0, abs(f_ref)), name
        # iteration 1 on its own, against the reference's arithmetic (serial oracle)
        x1 = x0.copy()
        st1 = mod.lm.LevenbergMarquardt(names[g["objective"]], hx(g["lambda"]), 10.0, 10.0, 1,
                                        0.0, solver=LM_CHOLESKY).minimize(x1)
        ser, xs, _, _ = O.lm_fd(oracle, names[g["objective"]], x0, lam=hx(g["lambda"]), max_iter=1,
                                f_delta=0.0, order=0)
        assert st1.iteration == ser.iteration == 1
        assert abs(st1.f_value - ser.f_value) <= 1e-8 * abs(ser.f_value), name
        if g["n"] <= 4:
            assert st1.f_value == ser.f_value and np.array_equal(x1, xs), name


def test_lm_default_functors_reject_qr_and_data(mod):
    from nlsolver_amd._capi import LM_QR, NlsgError
    with pytest.raises(NlsgError):
        mod.lm.LMEngine("rosenbrock", batch=1, n=4, solver=LM_QR)
    with pytest.raises(NlsgError):
        mod.lm.LMEngine(99, batch=1, n=4)
