"""Pins oracle/oracle_pso.c (serial restatement) to the reference's own PSO outputs."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle as O
from tests.test_oracle_golden import _fnv, hx

CASES = ["accel_2d_x0_3_3", "accel_256d_64p", "accel_8d_bounded", "accel_2d_default_stops",
         "accel_2d_other_coefs", "vanilla_16d_10p", "vanilla_16d_10p_bounded"]
X0 = {"accel_2d_x0_3_3": [3, 3], "accel_256d_64p": 0.3, "accel_8d_bounded": 2.0,
      "accel_2d_default_stops": [3, 3], "accel_2d_other_coefs": [3, 3],
      "vanilla_16d_10p": 2.0, "vanilla_16d_10p_bounded": 2.0}


@pytest.mark.parametrize("name", CASES)
def test_pso_serial_matches_reference(oracle, golden, name):
    g = golden("pso.json")[name]
    D, n = g["D"], g["particles"]
    x0 = X0[name]
    x = np.array(x0 if isinstance(x0, list) else [x0] * D, dtype=np.float64)
    lower = np.full(D, hx(g["lower"]))
    upper = np.full(D, hx(g["upper"]))
    gen = O.XorShift()
    oracle.orc_xorshift_init(C.byref(gen))
    cap = n * (g["max_iter"] + 1) if "eval_f" in g else 0
    lg = None
    if cap:
        lx, lf = np.zeros((cap, D)), np.zeros(cap)
        lg = O.EvalLog(lx.ctypes.data_as(O.pd), lf.ctypes.data_as(O.pd), cap, 0, D)
    st = oracle.orc_pso_serial(0, 1, 1 if g["type"] == "accelerated" else 0, g["bounded"],
                               x.ctypes.data_as(O.pd), D, lower.ctypes.data_as(O.pd),
                               upper.ctypes.data_as(O.pd), C.byref(gen), hx(g["inertia"]),
                               hx(g["cognitive"]), hx(g["social"]), n, g["max_iter"],
                               g["no_change"], hx(g["eps"]), C.byref(lg) if lg else None)
    assert (st.function_calls_used, st.iteration) == (g["fcalls"], g["iters"])
    assert st.f_value == hx(g["f"])
    assert x.tolist() == [hx(v) for v in g["x"]]
    after = [oracle.orc_xorshift_next(C.byref(gen)) for _ in range(2)]
    assert after == [hx(v) for v in g["rng_after"]]
    if lg:
        k = lg.count
        assert k == g["fcalls"]
        assert lf[:k].tolist() == [hx(v) for v in g["eval_f"]]
        assert _fnv(lx[:k]) == int(g["eval_x_fnv"])


def test_survey_anchor_values(golden):
    g = golden("pso.json")
    a = g["accel_2d_x0_3_3"]
    assert (a["fcalls"], hx(a["f"])) == (510, 0.00092613204254744167)
    b = g["accel_256d_64p"]
    assert hx(b["f"]) == 950.03762859425638


def test_deterministic_log_cos_within_one_ulp_of_libm(oracle):
    import math
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.uniform(0, 1, 20000), 2.0 ** -rng.uniform(0, 64, 5000)]):
        ref = math.log(x)
        assert abs(oracle.orc_log(x) - ref) <= 2.3e-16 * abs(ref) + 1e-300
    for y in rng.uniform(0, 6.283186, 20000):
        assert abs(oracle.orc_cos(y) - math.cos(y)) <= 1.2e-16
    assert oracle.orc_log(0.0) == -math.inf and oracle.orc_log(1.0) == 0.0
    assert oracle.orc_cos(0.0) == 1.0


def test_rnorm_table_logarithm_accuracy(oracle):
    """orc_log_unit (the table-driven logarithm of the normal variates, nlsg_math.h det_log_unit):
    within 0.62 ulp of the exact value on [2^-64, 1] (200-bit arithmetic), hence within 1 ulp of
    libm's log; exact at the ends; the normal variate itself is what libm's formula gives, to
    rounding."""
    import ctypes as C
    import math
    import mpmath as mp
    oracle.orc_log_unit.restype, oracle.orc_log_unit.argtypes = C.c_double, [C.c_double]
    mp.mp.prec = 200
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.random(6000), 1 - rng.random(6000) * 2.0 ** -rng.integers(1, 52, 6000),
                         rng.random(6000) * 2.0 ** -rng.integers(0, 63, 6000).astype(float),
                         [0.5, 2.0 ** -64, 0.70710678118654746, 0.70710678118654757, 1 - 2.0 ** -53]])
    worst = 0.0
    for x in xs:
        x = float(x)
        if not 2.0 ** -64 <= x < 1.0:
            continue
        v, e = oracle.orc_log_unit(x), mp.log(mp.mpf(x))
        worst = max(worst, float(abs(mp.mpf(v) - e) / mp.mpf(float(np.spacing(abs(float(e)))))))
    assert worst <= 0.62, worst
    assert oracle.orc_log_unit(1.0) == 0.0 and oracle.orc_log_unit(2.0 ** -64) == -64 * math.log(2.0)
    for x in rng.random(100000):
        ref = math.log(x)
        assert abs(oracle.orc_log_unit(x) - ref) <= np.spacing(abs(ref))
    for z in rng.integers(1, 2**64, size=20000, dtype=np.uint64):
        z = int(z)
        u1, u2 = z * 2.0 ** -64, (z & 0xFFFFFFFF) * 2.0 ** -32
        ref = math.sqrt(-2 * math.log(u1)) * math.cos(2 * 3.141593 * u2)
        assert abs(oracle.orc_rnorm(z) - ref) <= 6e-16 * max(1.0, abs(ref))


def test_rnorm_one_polynomial_cosine_accuracy(oracle):
    """orc_cos_unit (nlsg_math.h det_rnorm's cosine: one sine polynomial on [-pi/2, pi/2] after a
    reduction by pi): within 2.3e-16 (absolute) of libm's cos on the arguments that occur,
    [0, 2 * 3.141593], exact at 0."""
    import ctypes as C
    import math
    oracle.orc_cos_unit.restype, oracle.orc_cos_unit.argtypes = C.c_double, [C.c_double]
    rng = np.random.default_rng(11)
    ys = np.concatenate([rng.uniform(0, 6.283186, 100000),
                         [0.0, 6.283186, math.pi / 2, math.pi, 1.5 * math.pi, 2 * math.pi]])
    for y in ys:
        assert abs(oracle.orc_cos_unit(float(y)) - math.cos(float(y))) <= 2.3e-16
    assert oracle.orc_cos_unit(0.0) == 1.0 and oracle.orc_cos_unit(math.pi) == -1.0


def test_sync_pso_converges_like_the_reference(oracle, golden):
    """Accelerated PSO, Rosenbrock-2D, 10 particles, 50 iterations from bounds +-3: the
    synchronous counter-RNG algorithm reaches the reference's quality (f ~ 1e-3)."""
    ref = golden("pso.json")["accel_2d_x0_3_3"]
    fs = []
    for seed in range(8):
        run = O.PSOSyncRun(oracle, "rosenbrock", 10, 2, -3.0, 3.0, max_iter=50,
                           best_val_no_change=1000, seed=1000 + seed)
        run.step(60)
        assert run.s.done and run.s.iter == 50 and run.s.fevals == 510 == ref["fcalls"]
        fs.append(run.s.gbest_val)
    assert np.median(fs) < 20 * hx(ref["f"]) + 0.05 and min(fs) < 0.05
