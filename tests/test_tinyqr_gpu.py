"""Parity tests: batched tinyqr::lm on the device (nlsg_tinyqr_lm, tinyqr.h:461-470) vs
oracle_lm.c — bit for bit against the kernel-order restatement (order 1), to rounding against
the reference-pinned literal one (order 0) and against the reference's own outputs
(tests/golden/tinyqr.json, lm.json)."""
import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_lm_golden import RECT, _fnv, hx, rect_system

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tq():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd.tinyqr


def oracle_lm(oracle, X, y, order, tol=1e-12):
    p, n = X.shape
    beta = np.zeros(p)
    Xc = np.ascontiguousarray(X.reshape(-1))
    oracle.orc_tinyqr_lm_tol(O._ptr(Xc), O._ptr(np.ascontiguousarray(y)), n, p, O._ptr(beta), order, tol)
    return beta


def test_tinyqr_example_on_the_device(tq, oracle, golden):
    """X = [1 1 1 1; 0 1 2 3]^T, y = (1, 3, 5, 7.5) -> beta = (0.9, 2.15)."""
    X = np.array([[1, 1, 1, 1], [0, 1, 2, 3]], dtype=np.float64)
    y = np.array([1, 3, 5, 7.5])
    beta = tq.lm(X, y)
    assert np.array_equal(beta, oracle_lm(oracle, X, y, 1))
    ref = np.array([hx(v) for v in golden("lm.json")["tinyqr_example"]["beta"]])
    assert np.allclose(beta, ref, rtol=1e-13, atol=0) and np.allclose(beta, [0.9, 2.15])


@pytest.mark.parametrize("n,p,seed", RECT)
def test_reference_systems_on_the_device(tq, oracle, golden, n, p, seed):
    """The systems of tests/golden/tinyqr.json (outputs of the unmodified reference): the device
    equals the order-1 oracle bit for bit and the reference's beta to rounding."""
    Xf, y = rect_system(oracle, n, p, seed)
    X = Xf.reshape(p, n)
    beta = tq.lm(X, y)
    assert np.array_equal(beta, oracle_lm(oracle, X, y, 1)), (n, p)
    ref = np.array([hx(v) for v in golden("tinyqr.json")[f"rect_{n}x{p}"]["beta"]])
    assert np.allclose(beta, ref, rtol=1e-9, atol=1e-12)
    # the bound that belongs to the system, not a flat one: two backward-stable orthogonal
    # factorisations of the same X agree to cond(X) eps (here with a factor of 8 for n p rotations'
    # worth of rounding; measured: well inside)
    cond = np.linalg.cond(X.T)
    eps = np.finfo(np.float64).eps
    assert np.max(np.abs(beta - ref)) <= 8 * cond * eps * max(np.max(np.abs(ref)), 1e-300), (n, p, cond)


@pytest.mark.parametrize("n,p,batch", [(576, 64, 24), (64, 64, 40), (65, 64, 7), (130, 63, 9),
                                       (1000, 3, 33), (2, 1, 5), (1, 1, 3), (7, 7, 11), (33, 32, 6),
                                       (300, 17, 13), (129, 2, 4), (200, 62, 5), (64, 33, 4), (9, 8, 3), (40, 9, 3), (100, 31, 2), (8, 8, 2)])
def test_random_batches_bit_exact(tq, oracle, n, p, batch):
    """Batches of random systems incl. the augmented damped system of configs[3] (576 x 64), square
    ones, one column, one row: every system equals the order-1 oracle bit for bit, the literal
    (reference-pinned) oracle to rounding, and numpy's least-squares solution."""
    rng = np.random.default_rng(n * 1000 + p)
    X = 2 * rng.random((batch, p, n)) - 1
    y = 2 * rng.random((batch, n)) - 1
    beta = tq.lm(X, y)
    for b in range(batch):
        assert np.array_equal(beta[b], oracle_lm(oracle, X[b], y[b], 1)), (b, n, p)
    for b in range(min(batch, 4)):
        lit = oracle_lm(oracle, X[b], y[b], 0)
        ls = np.linalg.lstsq(X[b].T, y[b], rcond=None)[0]
        scale = np.max(np.abs(ls)) + 1e-300
        assert np.max(np.abs(beta[b] - lit)) <= 1e-9 * scale
        assert np.max(np.abs(beta[b] - ls)) <= 1e-7 * scale


def test_damped_normal_equations_agree_with_the_lm_engine_path(tq, oracle):
    """The square damped system the LM engine's QR solver handles inside its iteration (J^T J +
    lambda I, n = p = 64): the stand-alone entry point gives the oracle's bits too."""
    rng = np.random.default_rng(64)
    B = 2 * rng.random((512, 64)) - 1
    M = B.T @ B + 10.0 * np.eye(64)
    g = 2 * rng.random(64) - 1
    X = np.ascontiguousarray(M.T)  # column-major
    beta = tq.lm(X, g)
    assert np.array_equal(beta, oracle_lm(oracle, X, g, 1))
    assert np.allclose(M @ beta, g, rtol=1e-9, atol=1e-9)


def test_tolerance_and_rank_deficiency_follow_the_reference_rule(tq, oracle):
    """lm()'s cleanup reads |R| < tol as 0 (tinyqr.h:278-282): a duplicated column makes the last
    pivot a rounding remnant; with the default tol the division by the cleaned 0 gives inf / nan
    exactly as the restatement does, with tol = 0 the remnant is used."""
    rng = np.random.default_rng(5)
    X = 2 * rng.random((3, 40)) - 1
    X[2] = X[1]
    y = 2 * rng.random(40) - 1
    for tol in (1e-12, 0.0, 1e-3):
        beta = tq.lm(X, y, tol)
        ref = oracle_lm(oracle, X, y, 1, tol)
        assert np.array_equal(beta, ref, equal_nan=True), (tol, beta, ref)


def test_shape_errors(tq):
    import nlsolver_amd
    with pytest.raises(nlsolver_amd.NlsgError):
        tq.lm(np.zeros((65, 70)), np.zeros(70))   # p > 64
    with pytest.raises(nlsolver_amd.NlsgError):
        tq.lm(np.zeros((5, 3)), np.zeros(3))      # n < p
    with pytest.raises(TypeError):
        tq.lm(np.zeros((2, 5, 3)), np.zeros(3))


def test_full_size_property_residual_is_orthogonal(tq):
    """8192 systems of 576 x 64 (the batch of configs[3], augmented form): the least-squares
    residual is orthogonal to the columns — a size-independent property checked on every system."""
    rng = np.random.default_rng(8192)
    batch, n, p = 8192, 576, 64
    X = 2 * rng.random((batch, p, n)) - 1
    y = 2 * rng.random((batch, n)) - 1
    beta, ms = tq.lm(X, y, return_ms=True)
    r = y - np.einsum("bpn,bp->bn", X, beta)
    ortho = np.einsum("bpn,bn->bp", X, r)
    assert np.max(np.abs(ortho)) < 1e-9 * n
    assert ms > 0


# ---- reference-order mode (nlsg_tinyqr_qr): the reference's own bits on the device ----------------

@pytest.mark.parametrize("n,p,seed", RECT)
def test_reference_order_q_r_beta_equal_the_reference_bit_for_bit(tq, oracle, golden, n, p, seed):
    """tinyqr::qr_decomposition and lm on the device in the reference's order of operations
    (serial rotation order, Q formed, two products and an add per element, index-order sums):
    Q, R and beta equal the OUTPUTS OF THE UNMODIFIED REFERENCE (tests/golden/tinyqr.json: hashes
    of Q and R for every system, the matrices themselves for the small ones) bit for bit — 0 ulp
    where the co-rotated fast path agrees to rounding (rtol 1e-9 above)."""
    g = golden("tinyqr.json")[f"rect_{n}x{p}"]
    Xf, y = rect_system(oracle, n, p, seed)
    X = Xf.reshape(p, n)
    Q, R = tq.qr_decomposition(X, 1e-8)
    assert Q.shape == (p, n) and R.shape == (p, p)
    assert _fnv(Q.reshape(-1)) == int(g["Q_fnv"]) and _fnv(R.reshape(-1)) == int(g["R_fnv"])
    if "Q" in g:
        assert Q.reshape(-1).tolist() == [hx(v) for v in g["Q"]]
        assert R.reshape(-1).tolist() == [hx(v) for v in g["R"]]
    beta = tq.lm(X, y, reference_order=True)
    assert beta.tolist() == [hx(v) for v in g["beta"]]
    beta8 = tq.lm(X, y, 1e-8, reference_order=True)
    assert beta8.tolist() == [hx(v) for v in g["beta_tol1e-8"]]


@pytest.mark.parametrize("n,p,batch", [(64, 64, 9), (100, 70, 3), (300, 130, 2), (40, 9, 17), (1000, 200, 1)])
def test_reference_order_batches_vs_literal_oracle(tq, oracle, n, p, batch):
    """Batches, and column counts past the wavefront kernel's 64: Q and R equal the literal
    (reference-pinned, order 0) oracle bit for bit; beta too where it is offered (p <= 64)."""
    rng = np.random.default_rng(7 * n + p)
    X = 2 * rng.random((batch, p, n)) - 1
    y = 2 * rng.random((batch, n)) - 1
    Q, R = tq.qr_decomposition(X, 1e-8)
    for b in range(batch):
        Qo, Ro = np.zeros(n * p), np.zeros(p * p)
        oracle.orc_qr_decomposition(O._ptr(np.ascontiguousarray(X[b].reshape(-1))), n, p, 1e-8,
                                    O._ptr(Qo), O._ptr(Ro))
        assert np.array_equal(Q[b].reshape(-1), Qo) and np.array_equal(R[b].reshape(-1), Ro), b
    if p <= 64:
        beta = tq.lm(X, y, reference_order=True)
        for b in range(batch):
            assert np.array_equal(beta[b], oracle_lm(oracle, X[b], y[b], 0)), b


def test_reference_order_limits(tq):
    import nlsolver_amd
    with pytest.raises(nlsolver_amd.NlsgError):  # n + p beyond one wave's registers
        tq.qr_decomposition(np.zeros((10, 1300)))
    with pytest.raises(nlsolver_amd.NlsgError):  # n < p
        tq.qr_decomposition(np.zeros((5, 3)))
    with pytest.raises(nlsolver_amd.NlsgError):  # reference-order beta: p <= 64
        tq.lm(np.zeros((70, 100)), np.zeros(100), reference_order=True)
