"""Parity tests of the LM engine past 64 parameters (the reference class has no limit,
nlsolver.h:3428-3545): the workgroup-per-problem kernels (lm_wide_step_kernel: Cholesky in the
reference's per-element sums; lm_wide_tanh_eval_kernel: Gauss-Newton functors; lm_wide_fd_eval_kernel:
the default fin_diff / fin_diff_h functors) against oracle_lm.c in the kernels' operation order
(order = 1) — bit-exact parameters, objective values, damping and counters — and against the
reference's arithmetic (order = 0) within rounding."""
import numpy as np
import pytest

from tests import _oracle as O
from tests.test_lm_gpu import fd_starts, problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd


def check(st, ref, th, xr, lam, lam_r, tag):
    assert (st.iteration, st.function_calls_used, st.gradient_evals_used, st.hessian_evals_used) == \
           (ref.iteration, ref.function_calls_used, ref.gradient_evals_used, ref.hessian_evals_used), tag
    assert np.array_equal(st.f_value, ref.f_value, equal_nan=True), (tag, st.f_value, ref.f_value)
    assert np.array_equal(th, xr, equal_nan=True), (tag, np.flatnonzero(th != xr)[:8])
    assert np.array_equal(lam, lam_r, equal_nan=True), tag


@pytest.mark.parametrize("m,n,batch", [(80, 65, 3), (200, 100, 3), (300, 128, 2), (150, 129, 2),
                                       (260, 200, 2), (40, 70, 2)])
@pytest.mark.parametrize("kw", [dict(lam=10.0, max_iter=8, f_delta=0.0),
                                dict(lam=0.5, up=4.0, down=3.0, max_iter=100, f_delta=1e-12)])
def test_lm_wide_tanh_bit_exact_vs_kernel_order_oracle(mod, oracle, m, n, batch, kw):
    A, y, t0 = problems(oracle, 7, batch, m, n)
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        check(st[b], ref, th[b], xr, lam[b], lam_r, (m, n, b))


def test_lm_wide_tanh_agrees_with_reference_arithmetic(mod, oracle):
    """Against the serial oracle (the reference's sequential sums and libm tanh): after one
    iteration the parameters agree to rounding, and the run converges to the planted solution."""
    m, n = 400, 96
    A, y, t0 = problems(oracle, 3, 2, m, n)
    for b in range(2):
        x = t0[b].copy()
        st = mod.LevenbergMarquardt(mod.TanhRegression(A[b], y[b]), 10.0, 10.0, 10.0, 1, 0.0).minimize(x)
        ser, xs, _, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=0, lam=10.0, max_iter=1, f_delta=0.0)
        assert st.iteration == ser.iteration == 1
        assert np.max(np.abs(x - xs)) <= 1e-12 and abs(st.f_value - ser.f_value) <= 1e-12 * ser.f_value
        x = t0[b].copy()
        st = mod.LevenbergMarquardt(mod.TanhRegression(A[b], y[b]), 10.0, 10.0, 10.0, 30, 0.0).minimize(x)
        assert st.f_value < 1e-20


@pytest.mark.parametrize("objective,n,scale", [("rosenbrock", 65, 1.0), ("sphere", 100, 3.0),
                                               ("styblinski_tang", 129, 4.0), ("rastrigin", 70, 4.0),
                                               ("rosenbrock", 130, 1.0)])
def test_lm_wide_default_functors_bit_exact_vs_oracle(mod, oracle, objective, n, scale):
    kw = dict(lam=1.0, up=4.0, down=3.0, max_iter=2, f_delta=0.0)
    batch = 2
    x0 = fd_starts(oracle, objective, batch, n, scale)
    with mod.lm.LMEngine(objective, batch=batch, n=n, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_fd(oracle, objective, x0[b], order=1, **kw)
        check(st[b], ref, x[b], xr, lam[b], lam_r, (objective, n, b))
        assert st[b].function_calls_used == 3 * (1 + 4 * n + 16 * n * n)


def test_lm_wide_default_functors_long_point(mod, oracle):
    """A probe point past 256 coordinates (more than two 128-coordinate chunks per wave)."""
    n, kw = 300, dict(lam=1.0, max_iter=1, f_delta=0.0)
    x0 = fd_starts(oracle, "sphere", 1, n, 2.0)
    with mod.lm.LMEngine("sphere", batch=1, n=n, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    ref, xr, lam_r, _ = O.lm_fd(oracle, "sphere", x0[0], order=1, **kw)
    check(st[0], ref, x[0], xr, lam[0], lam_r, n)


def test_lm_wide_custom_objective_equals_builtin(mod):
    rosen = "double t1 = 1 - xi; double t2 = (xn - xi * xi); return t1 * t1 + 100 * t2 * t2;"
    n, kw = 72, dict(lam=1.0, max_iter=1, f_delta=0.0)
    x0 = np.linspace(-1.0, 1.0, 2 * n).reshape(2, n)
    out = []
    for obj in ("rosenbrock", mod.CustomObjective(rosen, chain=True)):
        with mod.lm.LMEngine(obj, batch=2, n=n, **kw) as eng:
            x, st, lam = eng.minimize(x0.copy())
        out.append((x, [(s.f_value, s.function_calls_used) for s in st], lam))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_lm_wide_whole_vector_objective_equals_builtin(mod):
    """A whole-vector user objective (x.sum over all coordinates: the lane-tree order of the
    built-in objectives) behind the default functors at n = 70: the bits of the built-in sphere."""
    n, kw = 70, dict(lam=1.0, max_iter=2, f_delta=0.0)
    x0 = np.linspace(-2.0, 2.0, 2 * n).reshape(2, n)
    out = []
    for obj in ("sphere", mod.CustomObjective(
            "return x.sum([](double xi, uint64_t) { return xi * xi; });", vector=True)):
        with mod.lm.LMEngine(obj, batch=2, n=n, **kw) as eng:
            x, st, lam = eng.minimize(x0.copy())
        out.append((x, [(s.f_value, s.function_calls_used) for s in st], lam))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


def test_lm_wide_diagonal_shortcut_and_limits(mod, oracle):
    """Sphere's finite-difference Hessian at a point where every off-diagonal probe difference
    vanishes takes is_diagonal's shortcut (:310-318) — same branch in kernel and oracle; QR and
    n > 1024 are refused."""
    from nlsolver_amd._capi import LM_QR, NlsgError
    n, kw = 66, dict(lam=2.0, max_iter=2, f_delta=0.0)
    x0 = np.zeros((1, n))
    with mod.lm.LMEngine("sphere", batch=1, n=n, **kw) as eng:
        x, st, lam = eng.minimize(x0.copy())
    ref, xr, lam_r, _ = O.lm_fd(oracle, "sphere", x0[0], order=1, **kw)
    check(st[0], ref, x[0], xr, lam[0], lam_r, "diag")
    A, y, _ = problems(oracle, 0, 1, 80, 70)
    with pytest.raises(NlsgError):
        mod.LMEngine(mod.TanhRegression(A, y), solver=LM_QR)
    with pytest.raises(NlsgError):
        mod.lm.LMEngine("sphere", batch=1, n=1025)


@pytest.mark.parametrize("m,n,batch", [(512, 128, 4), (33, 127, 2), (16, 65, 2), (100, 101, 3), (257, 112, 2),
                                       (64, 96, 2), (1, 66, 2),
                                       (200, 129, 2), (130, 257, 2), (96, 400, 1), (300, 256, 2), (50, 513, 1),
                                       (40, 144, 2), (64, 700, 1), (24, 1024, 1), (30, 1009, 1), (10, 150, 2), (1, 200, 1), (17, 255, 2)])
def test_lm_wide_matrix_core_kernel_bit_exact(mod, oracle, monkeypatch, m, n, batch):
    """n > 64 with J^T J on the matrix cores — the one-pass kernel up to 128 parameters
    (lm_wide128x8_tanh_eval_kernel), up to 256 (lm_wide256x8_tanh_eval_kernel), the super-block
    kernel beyond (lm_wide_mfma_tanh_eval_kernel: two, three and five column blocks, diagonal and
    off-diagonal passes), with the blocked matrix-core Cholesky step lm_wide_chol_step_kernel at every
    n > 64 (256, 512 and 1024 threads, odd n, a last block of one column, n = 1024) — against the
    order-1 oracle AND against the kernels they replace (NLSG_LM_WIDE_CHOL=0: the LDS-resident step
    lm_wide128_step_kernel up to n = 128, the column-wise one beyond; NLSG_LM_WIDE_MFMA=0: the VALU
    evaluations with the column-wise step): the benchmark size, odd n (scalar loads), m not a multiple of sixteen,
    fewer rows than a group, column blocks that are entirely padding. An fp64 MFMA is a k-ordered
    fma chain: same bits."""
    kw = dict(lam=10.0, max_iter=5, f_delta=0.0)
    A, y, t0 = problems(oracle, 11, batch, m, n)
    out = {}
    # default; the steps the blocked one replaced (LDS-resident up to n = 128, column-wise beyond)
    # under the same evaluations; VALU evaluations with the column-wise step
    for tag, env in (("1", {}), ("step", {"NLSG_LM_WIDE_CHOL": "0"}), ("0", {"NLSG_LM_WIDE_MFMA": "0"}),
                     ("sb", {"NLSG_LM_WIDE256": "0"})):  # (the super-block evaluation at 128 < n <= 256)
        for k in ("NLSG_LM_WIDE_CHOL", "NLSG_LM_WIDE_MFMA", "NLSG_LM_WIDE256"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
            out[tag] = eng.minimize(t0.copy())
    th, st, lam = out["1"]
    for b in range(batch):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        check(st[b], ref, th[b], xr, lam[b], lam_r, (m, n, b))
    for tag in ("step", "0", "sb"):
        assert np.array_equal(out[tag][0], th, equal_nan=True) and np.array_equal(out[tag][2], lam, equal_nan=True)
        assert [s.f_value for s in out[tag][1]] == [s.f_value for s in st]


def test_lm_wide_bench_configuration_sampled_parity(mod, oracle):
    """The configuration `bench.py --workload lm --lm-n 128` times — m = 512, n = 128, 1024 problems,
    20 iterations, lambda 10 — at its full batch: six sampled problems (first and last included)
    equal the order-1 oracle bit for bit, every problem converges to the planted solution."""
    m, n, batch = 512, 128, 1024
    kw = dict(lam=10.0, max_iter=20, f_delta=0.0)
    A, y, t0 = problems(oracle, 5, batch, m, n)
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in (0, 1, 257, 511, 1000, batch - 1):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        check(st[b], ref, th[b], xr, lam[b], lam_r, b)
    assert all(s.iteration == 20 and s.done == 1 for s in st)
    assert max(s.f_value for s in st) < 1e-20


def test_lm_wide_blocked_step_every_width(mod, oracle):
    """Every n from 65 to 176 (all positions of the last, partial sixteen-column block; odd and even
    row strides; two to eleven panels) through the blocked step and the matrix-core evaluations:
    each problem equals the order-1 oracle bit for bit after three iterations."""
    kw = dict(lam=2.0, max_iter=3, f_delta=0.0)
    for n in range(65, 177):
        m = 24 + (n % 5) * 7
        A, y, t0 = problems(oracle, 100 + n, 1, m, n)
        with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
            th, st, lam = eng.minimize(t0.copy())
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[0], y[0], t0[0], order=1, **kw)
        check(st[0], ref, th[0], xr, lam[0], lam_r, (m, n))


@pytest.mark.parametrize("n", [70, 130, 300])
def test_lm_wide_diagonal_hessian_takes_the_shortcut_on_the_evaluations_verdict(mod, oracle, n):
    """Design matrices with one non-zero per row make J^T J diagonal: get_update_with_hessian's
    shortcut (nlsolver.h:310-318) must fire, and past 64 parameters it does so on the verdict the
    evaluation kernel leaves for the step (one-pass kernels up to 256, super-blocks beyond) — as in
    the oracle; a second problem of the batch keeps a dense matrix, so both branches run side by side."""
    m, kw = 2 * n, dict(lam=2.0, max_iter=4, f_delta=0.0)
    A, y, t0 = problems(oracle, 900 + n, 2, m, n)
    rng = np.random.default_rng(n)
    D = np.zeros((m, n))
    D[np.arange(m), np.arange(m) % n] = 0.5 + rng.random(m)
    A[0] = D
    y[0] = np.tanh(D @ (0.3 * (2 * rng.random(n) - 1)))
    with mod.LMEngine(mod.TanhRegression(A, y), **kw) as eng:
        th, st, lam = eng.minimize(t0.copy())
    for b in range(2):
        ref, xr, lam_r, _ = O.lm_solve(oracle, A[b], y[b], t0[b], order=1, **kw)
        check(st[b], ref, th[b], xr, lam[b], lam_r, (n, b))
    # the diagonal problem really is one: a dense solve would have mixed the coordinates
    H = 2 * (A[0].T @ A[0])
    assert np.count_nonzero(H - np.diag(np.diag(H))) == 0
