"""Pins oracle/oracle_lm.c to the reference: LevenbergMarquardt with Gauss-Newton functors,
math::cholesky solve, tinyqr QR / lm."""
import math

import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import _fnv, hx

SEED = 12374563468
TANH = {"tanh_m16_n4": (0, 16, 4), "tanh_m64_n8_p3": (3, 64, 8), "tanh_m128_n64": (1, 128, 64),
        "tanh_m512_n64": (0, 512, 64), "tanh_m512_n64_default_stop": (5, 512, 64)}


def exp_model():
    t = np.array([0.25 * i for i in range(8)])
    y = np.array([2.0 * math.exp(-0.7 * ti) + 0.02 * math.sin(3.0 * i + 1.0) for i, ti in enumerate(t)])
    return t, y


@pytest.mark.parametrize("name", ["exp_default", "exp_lambda1_5iters"])
def test_lm_exp_model_matches_reference(oracle, golden, name):
    g = golden("lm.json")[name]
    t, y = exp_model()
    st, x, lam, flog = O.lm_solve(oracle, None, y, [1.0, -0.1], kind=0, t=t, lam=hx(g["lambda"]),
                                  max_iter=g["max_iter"], f_delta=hx(g["f_delta"]))
    assert (st.iteration, st.function_calls_used, st.gradient_evals_used, st.hessian_evals_used) == \
        (g["iters"], g["fcalls"], g["gcalls"], g["hcalls"])
    assert st.f_value == hx(g["f"]) and x.tolist() == [hx(v) for v in g["x"]]
    assert flog.tolist() == [hx(v) for v in g["f_vals"]]


@pytest.mark.parametrize("name", sorted(TANH))
def test_lm_tanh_regression_matches_reference(oracle, golden, name):
    g = golden("lm.json")[name]
    prob, m, n = TANH[name]
    A, y, th0 = O.tanh_problem(oracle, SEED, prob, m, n)
    st, x, lam, flog = O.lm_solve(oracle, A, y, th0, lam=hx(g["lambda"]), max_iter=g["max_iter"],
                                  f_delta=hx(g["f_delta"]))
    assert (st.iteration, st.function_calls_used) == (g["iters"], g["fcalls"])
    assert flog.tolist() == [hx(v) for v in g["f_vals"]]  # every objective value of the run
    assert x.tolist() == [hx(v) for v in g["x"]] and st.f_value == hx(g["f"])


def test_tinyqr_example_matches_reference(oracle, golden):
    g = golden("lm.json")["tinyqr_example"]
    X = np.array([1, 1, 1, 1, 0, 1, 2, 3], dtype=np.float64)  # column-major 4 x 2
    Q, R, beta = np.zeros(8), np.zeros(4), np.zeros(2)
    oracle.orc_qr_decomposition(O._ptr(X), 4, 2, 1e-8, O._ptr(Q), O._ptr(R))
    assert Q.tolist() == [hx(v) for v in g["Q"]] and R.tolist() == [hx(v) for v in g["R"]]
    yv = np.array([1, 3, 5, 7.5])
    oracle.orc_tinyqr_lm(O._ptr(X), O._ptr(yv), 4, 2, O._ptr(beta))
    assert beta.tolist() == [hx(v) for v in g["beta"]]
    assert np.allclose(beta, [0.9, 2.15])


@pytest.mark.parametrize("name,n,seed,lam", [("linalg_n4", 4, 1, 0.5), ("linalg_n16", 16, 2, 10.0),
                                             ("linalg_n64", 64, 3, 10.0)])
def test_cholesky_and_qr_solves_match_reference(oracle, golden, name, n, seed, lam):
    g = golden("lm.json")[name]
    k = oracle.orc_ctr_key(seed, 77)
    B = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, e)) - 1 for e in range(n * n)]).reshape(n, n)
    b = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, n * n + i)) - 1 for i in range(n)])
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            acc = 0.0
            for l in range(n):
                acc += B[l, i] * B[l, j]
            M[i, j] = acc + (lam if i == j else 0.0)
    upd, Mc = np.zeros(n), M.copy()
    oracle.orc_update_with_hessian(O._ptr(upd), O._ptr(Mc), O._ptr(b), n)
    assert upd.tolist() == [hx(v) for v in g["cholesky_solution"]]
    X = np.ascontiguousarray(M.T.reshape(-1))  # X[j*n+i] = M[i][j]
    beta = np.zeros(n)
    oracle.orc_tinyqr_lm(O._ptr(X), O._ptr(b), n, n, O._ptr(beta))
    assert beta.tolist() == [hx(v) for v in g["qr_solution"]]
    Q, R = np.zeros(n * n), np.zeros(n * n)
    oracle.orc_qr_decomposition(O._ptr(X), n, n, 1e-8, O._ptr(Q), O._ptr(R))
    assert _fnv(Q) == int(g["Q_fnv"]) and _fnv(R) == int(g["R_fnv"])
    assert np.allclose(upd, beta, rtol=1e-9)  # the two solves agree to rounding, not bitwise


def test_deterministic_exp_tanh_accuracy(oracle):
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.uniform(-30, 30, 20000), rng.uniform(-1, 1, 5000)]):
        assert abs(oracle.orc_exp(x) - math.exp(x)) <= 2.3e-16 * math.exp(x)
        assert abs(oracle.orc_tanh(x) - math.tanh(x)) <= 2.3e-16
    assert oracle.orc_tanh(0.0) == 0.0 and oracle.orc_tanh(40.0) == 1.0 and oracle.orc_exp(0.0) == 1.0


def test_deterministic_tanh_rational_form_properties(oracle):
    """orc_tanh = ((s - 1) B + 2 s r) / ((s + 1) B + 2 s r) (one division, oracle_lm.c): odd,
    non-decreasing, relatively accurate for small arguments (1 - 2 / (e + 1) is not: it cancels),
    exactly 1 from 22 on, NaN for NaN."""
    rng = np.random.default_rng(9)
    xs = np.sort(np.concatenate([rng.uniform(0, 23, 20000), 10.0 ** rng.uniform(-300, 0, 5000)]))
    vals = np.array([oracle.orc_tanh(float(x)) for x in xs])
    assert np.all(np.diff(vals) >= -2.3e-16)           # monotone up to the last bit
    for x, v in zip(xs[::7], vals[::7]):
        assert oracle.orc_tanh(-float(x)) == -v
        ref = math.tanh(float(x))
        assert abs(v - ref) <= 3 * np.spacing(ref)       # RELATIVE accuracy, small x included
    assert oracle.orc_tanh(22.0) == 1.0 and oracle.orc_tanh(1e6) == 1.0 and oracle.orc_tanh(18.0) < 1.0
    assert oracle.orc_tanh(1e-300) == 1e-300 and math.isnan(oracle.orc_tanh(float("nan")))
    # the sign comes from the argument's sign bit, zero included (libm: tanh(-0.0) = -0.0)
    assert math.copysign(1.0, oracle.orc_tanh(-0.0)) == -1.0 and oracle.orc_tanh(-0.0) == 0.0
    assert math.copysign(1.0, oracle.orc_tanh(0.0)) == 1.0
    assert oracle.orc_tanh(float("-inf")) == -1.0 and oracle.orc_tanh(float("inf")) == 1.0


@pytest.mark.parametrize("name", ["tanh_m128_n64", "tanh_m512_n64"])
@pytest.mark.parametrize("solver", [0, 1])
def test_kernel_order_and_qr_solver_agree_with_reference_arithmetic(oracle, golden, name, solver):
    """order=1 (kernel summation, fma chains, deterministic tanh) and the tinyqr damped solve
    follow the same path as the reference run: objective values within 1e-9 relative while
    they are above the rounding floor."""
    g = golden("lm.json")[name]
    prob, m, n = TANH[name]
    A, y, th0 = O.tanh_problem(oracle, SEED, prob, m, n)
    st, x, lam, flog = O.lm_solve(oracle, A, y, th0, max_iter=g["max_iter"], f_delta=0.0,
                                  solver=solver, order=1)
    ref = np.array([hx(v) for v in g["f_vals"]])
    assert st.iteration == g["iters"]
    big = ref > 1e-10  # well above the rounding floor of sum r^2 (~1e-30 .. 1e-16 near theta*)
    assert np.allclose(flog[big], ref[big], rtol=1e-9, atol=0)
    assert np.all(flog[~big] < 1e-9)
    assert flog[-1] < 1e-20 or ref[-1] > 1e-20  # both converge to the zero-residual solution


RECT = [(1, 1, 1), (3, 1, 2), (5, 3, 3), (12, 5, 4), (8, 8, 5), (65, 64, 6), (200, 17, 7), (576, 64, 8)]


def rect_system(oracle, n, p, seed):
    """The pseudo-random rectangular system of ref_driver's tinyqr-rect: X column-major n x p."""
    k = oracle.orc_ctr_key(seed, 78)
    X = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, e)) - 1 for e in range(n * p)])
    y = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, n * p + i)) - 1 for i in range(n)])
    return X, y


@pytest.mark.parametrize("n,p,seed", RECT)
def test_tinyqr_rectangular_systems_match_reference(oracle, golden, n, p, seed):
    """tinyqr::qr_decomposition / lm on n x p systems with n >= p (tinyqr.h:291-310, 437-470; SURVEY
    row a25): the restatement reproduces the reference's Q, R and beta bit for bit; the kernel
    arithmetic (order 1: co-rotated right-hand side, fma element updates) agrees to rounding."""
    g = golden("tinyqr.json")[f"rect_{n}x{p}"]
    X, y = rect_system(oracle, n, p, seed)
    beta = np.zeros(p)
    oracle.orc_tinyqr_lm(O._ptr(X), O._ptr(y), n, p, O._ptr(beta))
    assert beta.tolist() == [hx(v) for v in g["beta"]]
    Q, R = np.zeros(n * p), np.zeros(p * p)
    oracle.orc_qr_decomposition(O._ptr(X), n, p, 1e-8, O._ptr(Q), O._ptr(R))
    assert _fnv(Q) == int(g["Q_fnv"]) and _fnv(R) == int(g["R_fnv"])
    if "Q" in g:
        assert Q.tolist() == [hx(v) for v in g["Q"]] and R.tolist() == [hx(v) for v in g["R"]]
    b1 = np.zeros(p)
    oracle.orc_tinyqr_lm_order(O._ptr(X), O._ptr(y), n, p, O._ptr(b1), 1)
    assert np.allclose(b1, beta, rtol=1e-9, atol=1e-12)
    # and both are the least-squares solution
    ls = np.linalg.lstsq(X.reshape(p, n).T, y, rcond=None)[0]
    assert np.allclose(beta, ls, rtol=1e-8, atol=1e-10)
