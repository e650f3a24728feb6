"""The drop-in C++ header (include/nlsolver_mi/nlsolver.h), driven from user-style
C++ programs in tests/cpp/ (built with g++ -std=c++17)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from tests import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin")
LIB = os.path.join(ROOT, "nlsolver_amd", "libnlsolver_hip.so")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])
    return BIN


def hx(v):
    return float.fromhex(v)


def test_config_c1_through_header_matches_reference_bit_exact(built, golden):
    out = json.loads(subprocess.check_output([os.path.join(built, "header_c1")], text=True))
    gold = golden("de_c1.json")
    for name in ("c1_random_pop40_x0_5_7", "random_pop50_x0_5_7", "example_best_pop50_x0_2_7"):
        g, o = gold[name], out[name]
        assert (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"]), name
        assert hx(o["f"]) == hx(g["f"]) and [hx(v) for v in o["x"]] == [hx(v) for v in g["x"]]
        # the caller's generator advanced exactly as under the reference
        assert [hx(v) for v in o["rng_after"]] == [hx(v) for v in g["rng_after"]]
        assert o["grad"] == 0 and o["hess"] == 0
    g, o = gold["readme_objective_pop40"], out["readme_objective_pop40"]
    assert (o["fcalls"], o["iters"], hx(o["f"])) == (g["fcalls"], g["iters"], hx(g["f"]))
    assert [hx(v) for v in o["x"]] == [hx(v) for v in g["x"]]
    # lambda / const-ref functors ran and converged
    assert abs(hx(out["lambda"]["x"][0]) - 1) < 0.05 and abs(hx(out["lambda"]["x"][1])) < 0.05
    assert hx(out["const_ref_sphere"]["f"]) < 1e-3


def test_device_objective_has_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(built, "header_device")
    args = [exe, "random", "2", "40", "100", "1e-3", "50", "5"]
    r = subprocess.run(args, env=dict(os.environ, NLSG_LIBRARY=LIB), capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout
    r = subprocess.run(args, env=dict(os.environ, NLSG_LIBRARY="/nonexistent/lib.so"),
                       capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("strategy,D,pop,max_iter,eps,no_change,x0", [
    ("random", 2, 40, 1000, 10e-4, 50, 5.0),
    ("best", 2, 50, 1000, 10e-4, 50, 3.0),
    ("random", 128, 4096, 25, 0.0, 1000, 4.096),
    ("best", 130, 512, 10, 0.0, 1000, 2.0),
])
@pytest.mark.parametrize("kind", ["builtin", "custom"])
def test_device_objective_through_header_matches_oracle(built, oracle, strategy, D, pop, max_iter,
                                                        eps, no_change, x0, kind):
    """kind = custom: the objective is device::Custom<double> (source text compiled at solve time);
    it spells the Rosenbrock chain, so the same oracle run must come out."""
    # the test binary links no HIP runtime of its own: libnlsolver_hip.so brings the system one,
    # and with it the system hiprtc (the default of nlsg_rtc_load)
    env = dict(os.environ, NLSG_LIBRARY=LIB)
    out = subprocess.check_output(
        [os.path.join(built, "header_device"), strategy, str(D), str(pop), str(max_iter),
         repr(eps), str(no_change), repr(x0)] + (["custom"] if kind == "custom" else []),
        env=env, text=True)
    o = json.loads(out)
    assert "device_error" not in o, o
    # the header keys the device RNG with two draws of the caller's generator
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    half = [min(int(oracle.orc_xorshift_next(C.byref(xs)) * 2.0**32), 2**32 - 1) for _ in range(2)]
    seed = (half[0] << 32) | half[1]
    ref = O.DESyncRun(oracle, "rosenbrock", pop, D, [x0] * D, strategy=1 if strategy == "random" else 0,
                      eps=eps, max_iter=max_iter, best_val_no_change=no_change, seed=seed)
    while not ref.s.done:
        ref.step()
    assert (o["fcalls"], o["iters"]) == (ref.s.fcalls, ref.s.iter)
    assert hx(o["f"]) == ref.scores[ref.s.best_id]
    assert np.array_equal(np.array([hx(v) for v in o["x"]]), ref.best_x)
    after = [oracle.orc_xorshift_next(C.byref(xs)) for _ in range(2)]
    assert [hx(v) for v in o["rng_after"]] == after  # generator advanced by exactly two draws
    # host evaluation of the tagged objective agrees with the device value (1e-12)
    assert abs(hx(o["f_host"]) - hx(o["f"])) <= 1e-12 * max(1.0, abs(hx(o["f"])))
    if D == 2:
        assert np.all(np.abs(ref.best_x - 1.0) <= 0.05)  # the reference's own pass criterion


def test_pso_host_path_through_header_matches_reference_bit_exact(built, golden):
    out = json.loads(subprocess.check_output([os.path.join(built, "header_pso"), "host"], text=True))
    gold = golden("pso.json")
    for name in ("accel_2d_x0_3_3", "accel_256d_64p", "accel_8d_bounded", "accel_2d_default_stops"):
        g, o = gold[name], out[name]
        assert (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"]), name
        assert hx(o["f"]) == hx(g["f"]) and o["x"] == g["x"] and o["rng_after"] == g["rng_after"]
    # Vanilla with the intended update converges on the 2-D example (reference: UB)
    assert hx(out["vanilla_2d_intended_update"]["f"]) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,n,max_iter,eps,no_change,bound,bounded", [
    ("accel", 2, 10, 50, 0.0, 1000, 3.0, 0),
    ("accel", 256, 512, 12, 0.0, 1000, 2.048, 1),
    ("vanilla", 16, 64, 40, 1e-3, 50, 2.0, 1),
])
@pytest.mark.parametrize("objective", ["builtin", "custom"])
def test_pso_device_objective_through_header_matches_oracle(built, oracle, kind, D, n, max_iter,
                                                            eps, no_change, bound, bounded,
                                                            objective):
    out = subprocess.check_output(
        [os.path.join(built, "header_pso"), "device", kind, str(D), str(n), str(max_iter),
         repr(eps), str(no_change), repr(bound), str(bounded)] +
        (["custom"] if objective == "custom" else []),
        env=dict(os.environ, NLSG_LIBRARY=LIB), text=True)
    o = json.loads(out)
    assert "device_error" not in o, o
    o = o["run"]
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    half = [min(int(oracle.orc_xorshift_next(C.byref(xs)) * 2.0**32), 2**32 - 1) for _ in range(2)]
    seed = (half[0] << 32) | half[1]
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, -bound, bound, bounded=bool(bounded),
                       type=O.PSO_ACCELERATED if kind == "accel" else O.PSO_VANILLA, eps=eps,
                       max_iter=max_iter, best_val_no_change=no_change, seed=seed)
    while not ref.s.done:
        ref.step()
    assert (o["fcalls"], o["iters"]) == (ref.s.fevals, ref.s.iter)
    assert hx(o["f"]) == ref.s.gbest_val
    assert np.array_equal(np.array([hx(v) for v in o["x"]]), ref.gbest_x)
    after = [oracle.orc_xorshift_next(C.byref(xs)) for _ in range(2)]
    assert [hx(v) for v in o["rng_after"]] == after


@pytest.mark.parametrize("name", ["n8", "n64", "n1024", "n64_default_stop", "n100_ragged_start",
                                  "n130_alpha_half", "n256_max_iter_5"])
def test_bfgs_host_path_through_header_matches_reference_bit_exact(built, golden, name):
    g = golden("bfgs.json")[name]
    out = subprocess.check_output(
        [os.path.join(built, "header_bfgs"), "host", str(g["n"]), str(g["max_iter"]),
         repr(hx(g["grad_eps"])), repr(hx(g["alpha"])), repr(hx(g["x0"])), repr(hx(g["x0_step"]))],
        text=True)
    o = json.loads(out)
    assert (o["fcalls"], o["iters"], o["gcalls"]) == (g["fcalls"], g["iters"], g["gcalls"])
    assert o["f"] == g["f"] and o["x"][:8] == g["x_head"]


def test_bfgs_default_finite_difference_gradient_runs(built):
    o = json.loads(subprocess.check_output([os.path.join(built, "header_bfgs"), "findiff"], text=True))
    x = [hx(v) for v in o["x"]]
    assert abs(x[0] - 1) < 0.05 and abs(x[1] - 1) < 0.05  # example.cpp's BFGS on Rosenbrock
    assert o["fcalls"] > 4 * o["gcalls"]  # fin_diff counts its 4 evaluations per dimension


@pytest.mark.gpu
@pytest.mark.parametrize("summation,tree", [("reference", 0), ("tree", 1)])
def test_bfgs_device_batch_through_header_matches_oracle(built, oracle, summation, tree):
    """minimize_batch() of the quadratic: in reference order (the default) equal to the serial oracle —
    the restatement pinned to the reference's runs —, with NLSG_SUMMATION=tree to the tree oracle."""
    n, B, max_iter, grad_eps, alpha = 96, 5, 60, 1e-8, 1.0
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    if summation == "tree":
        env["NLSG_SUMMATION"] = "tree"
    out = subprocess.check_output(
        [os.path.join(built, "header_bfgs"), "device", str(n), str(B), str(max_iter),
         repr(grad_eps), repr(alpha)], env=dict(env, NLSG_LIBRARY=LIB), text=True)
    res = json.loads(out)
    assert isinstance(res, list) and len(res) == B, res
    import math
    for p, o in enumerate(res):
        x0 = np.array([1.0 + 0.01 * p * math.cos(0.3 * i) for i in range(n)])
        ref, xr, _ = O.bfgs_quad(oracle, x0, max_iter=max_iter, grad_eps=grad_eps, alpha=alpha, tree=tree)
        assert (o["fcalls"], o["iters"], o["gcalls"]) == \
            (ref.function_calls_used, ref.iteration, ref.gradient_evals_used)
        assert hx(o["f"]) == ref.f_value
        assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device-fd", "device-fd-custom"])
@pytest.mark.parametrize("n", [2, 16, 128])
def test_bfgs_default_gradient_on_device_objective_through_header(built, oracle, n, mode):
    """BFGS<device::Rosenbrock<double>>(f).minimize(x): the default fin_diff gradient runs on the
    GPU (nlsolver.h:1385-1413 restated in the search kernel); with NLSG_SUMMATION=tree (the throughput
    kernels; a Custom objective always) bit-exact vs the tree oracle."""
    args = dict(max_iter=8, grad_eps=0.0, alpha=1.0)
    out = subprocess.check_output(
        [os.path.join(built, "header_bfgs"), mode, str(n), "8", "0.0", "1.0", "0.9", "0.001"],
        env=dict(os.environ, NLSG_LIBRARY=LIB, NLSG_SUMMATION="tree"), text=True)
    o = json.loads(out)
    assert "device_error" not in o, o
    x0 = 0.9 + 0.001 * np.arange(n, dtype=np.float64)
    ref, xr, _, _ = O.bfgs_fd(oracle, "rosenbrock", x0, tree=1, **args)
    assert (o["fcalls"], o["iters"], o["gcalls"]) == \
        (ref.function_calls_used, ref.iteration, ref.gradient_evals_used)
    assert hx(o["f"]) == ref.f_value
    assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


OBJECTIVE_NAMES = {0: "rosenbrock", 1: "sphere", 2: "styblinski_tang"}


def _fnv(x):
    h = 1469598103934665603  # FNV-1a over the bytes of x, as the reference driver hashed it
    for byte in np.ascontiguousarray(np.asarray(x, dtype=np.float64)).view(np.uint8).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _bfgs_fd_args(g):
    return [str(g["n"]), str(g["max_iter"]), repr(hx(g["grad_eps"])), repr(hx(g["alpha"])),
            repr(hx(g["x0"])), repr(hx(g["x0_step"]))]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rosenbrock_n2", "rosenbrock_n4", "rosenbrock_n16_default_stop",
                                  "rosenbrock_n128_20iters", "sphere_n5", "sphere_n130_alpha_half",
                                  "styblinski_tang_n8"])
def test_bfgs_minimize_through_header_is_the_reference_run(built, golden, name):
    """The drop-in's default: BFGS<device::Rosenbrock<double>, double>(f, ...).minimize(x) — the
    reference's own call with the objective type swapped — solves in reference order
    (device::summation()) and returns the reference's run (tests/golden/bfgs_fd.json,
    made by the unmodified reference): every count, f and x BIT FOR BIT."""
    g = golden("bfgs_fd.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_bfgs"), "device-fd", *_bfgs_fd_args(g), OBJECTIVE_NAMES[g["objective"]]],
        env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in o, o
    assert (o["fcalls"], o["iters"], o["gcalls"]) == (g["fcalls"], g["iters"], g["gcalls"])
    assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rosenbrock_n2", "rosenbrock_n4", "rosenbrock_n16_default_stop",
                                  "rosenbrock_n128_20iters"])
def test_bfgs_custom_terms_through_header_run_in_index_order(built, golden, name):
    """A Custom objective given by its terms (here the Rosenbrock chain as source text) is summed in
    index order by default — what the body's own loop on a CPU does — so the same text as the
    reference's functor gives the reference's run bit for bit."""
    g = golden("bfgs_fd.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_bfgs"), "device-fd-custom", *_bfgs_fd_args(g)],
        env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in o, o
    assert (o["fcalls"], o["iters"], o["gcalls"]) == (g["fcalls"], g["iters"], g["gcalls"])
    assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n8", "n64", "n64_default_stop", "n100_ragged_start", "n130_alpha_half",
                                  "n256_max_iter_5", "n1024"])
def test_bfgs_quadratic_minimize_through_header_is_the_reference_run(built, golden, name):
    """BFGS<device::QuadDiagRank1<double>, double>(f, ...).minimize(x): the reference's runs on the G6
    quadratic with its analytic gradient functor (tests/golden/bfgs.json) bit for bit, n = 1024 —
    BASELINE configs[2]'s dimension — included."""
    g = golden("bfgs.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_bfgs"), "device-one", *_bfgs_fd_args(g)],
        env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in o, o
    assert (o["fcalls"], o["iters"], o["gcalls"]) == (g["fcalls"], g["iters"], g["gcalls"])
    assert o["f"] == g["f"] and o["x"][:8] == g["x_head"]
    assert _fnv([hx(v) for v in o["x"]]) == int(g["x_fnv"])


@pytest.mark.gpu
def test_bfgs_summation_switch_through_header(built, oracle, golden):
    """minimize() and minimize_batch() solve in reference order unless NLSG_SUMMATION=tree; a misspelt
    value is an error, not a default."""
    g = golden("bfgs_fd.json")["rosenbrock_n16_default_stop"]
    cmd = [os.path.join(built, "header_bfgs"), "device-fd", *_bfgs_fd_args(g), "rosenbrock"]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    env["NLSG_LIBRARY"] = LIB
    x0 = hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)
    tree, xt, _, _ = O.bfgs_fd(oracle, "rosenbrock", x0, tree=1, max_iter=g["max_iter"],
                               grad_eps=hx(g["grad_eps"]), alpha=hx(g["alpha"]))

    def is_tree(o):
        return hx(o["f"]) == tree.f_value and np.array_equal(np.array([hx(v) for v in o["x"]]), xt) and \
            (o["fcalls"], o["iters"], o["gcalls"]) == \
            (tree.function_calls_used, tree.iteration, tree.gradient_evals_used)

    def is_reference(o):
        return o["f"] == g["f"] and o["x"] == g["x"] and (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"])

    run = lambda extra, **kw: json.loads(subprocess.check_output(cmd + extra, env=dict(env, **kw), text=True))
    assert tree.f_value != hx(g["f"])  # (the two orders do differ on this run)
    assert is_reference(run(["batch"]))
    assert is_reference(run(["batch"], NLSG_SUMMATION="reference"))
    assert is_tree(run(["batch"], NLSG_SUMMATION="tree"))
    assert is_tree(run([], NLSG_SUMMATION="tree"))
    assert is_reference(run([]))
    r = subprocess.run(cmd, env=dict(env, NLSG_SUMMATION="refrence"), capture_output=True, text=True)
    assert r.returncode == 3 and "NLSG_SUMMATION" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_summation_switch_rejects_a_misspelt_value_without_a_gpu(built):
    """NLSG_SUMMATION is read before anything touches the device: a value that is neither `reference`
    nor `tree` is an error (no silent default), here with the real library and no GPU in sight."""
    for prog, args in (("header_bfgs", ["device-fd", "4", "5", "0.0", "1.0", "0.9", "0.01"]),
                       ("header_nm_lm", ["lm-device-fd", "4", "10", "5", "0.0", "0.8", "0.01"])):
        r = subprocess.run([os.path.join(built, prog), *args],
                           env=dict(os.environ, NLSG_LIBRARY=LIB, NLSG_SUMMATION="automatic"),
                           capture_output=True, text=True)
        assert r.returncode == 3 and "NLSG_SUMMATION must be reference or tree" in r.stdout, (r.stdout, r.stderr)


def test_bfgs_device_objective_without_library_fails_loudly(built):
    """No CPU fallback on the device path: without the HIP library the call throws."""
    r = subprocess.run([os.path.join(built, "header_bfgs"), "device-fd", "4", "5", "0.0", "1.0",
                        "0.9", "0.01"], env=dict(os.environ, NLSG_LIBRARY="/nonexistent/lib.so"),
                       capture_output=True, text=True)
    assert r.returncode == 3 and "device_error" in r.stdout, (r.returncode, r.stdout, r.stderr)


NM_CASES = ["example_2d", "d4_200iters", "d4_fixed_step", "d16_bounded", "d8_restarts",
            "d6_maximize_bounded", "d130_ragged"]


def _nm_args(g):
    return [str(g["D"]), repr(hx(g["step"])), str(g["max_iter"]), repr(hx(g["eps"])),
            str(g["no_change"]), str(g["restarts"]), repr(hx(g["x0"])), repr(hx(g["x0_step"])),
            str(g["bounded"]), repr(hx(g["upper"])), repr(hx(g["lower"])), str(g["minimize"])]


@pytest.mark.parametrize("name", NM_CASES)
def test_nm_host_path_through_header_matches_reference_bit_exact(built, golden, name):
    g = golden("nm.json")[name]
    o = json.loads(subprocess.check_output([os.path.join(built, "header_nm_lm"), "nm-host",
                                            *_nm_args(g)], text=True))
    assert (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"])
    assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.parametrize("name,args", [("exp_default", []), ("exp_lambda1_5iters", ["1", "5", "0"])])
def test_lm_host_path_through_header_matches_reference_bit_exact(built, golden, name, args):
    g = golden("lm.json")[name]
    o = json.loads(subprocess.check_output([os.path.join(built, "header_nm_lm"), "lm-host-exp",
                                            *args], text=True))
    assert (o["fcalls"], o["iters"], o["gcalls"], o["hcalls"]) == \
        (g["fcalls"], g["iters"], g["gcalls"], g["hcalls"])
    assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["example_2d", "d4_200iters", "d16_bounded", "d8_restarts"])
@pytest.mark.parametrize("mode", ["nm-device", "nm-device-custom"])
def test_nm_device_objective_through_header_matches_oracle(built, oracle, golden, name, mode):
    """NelderMead<device::Rosenbrock<double>, double>(f, ...).minimize(x): in reference order (the default)
    the reference's own run (tests/golden/nm.json) bit for bit; with NLSG_SUMMATION=tree the tree oracle."""
    g = golden("nm.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    ref_run = json.loads(subprocess.check_output([os.path.join(built, "header_nm_lm"), mode, *_nm_args(g)],
                                                 env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in ref_run, ref_run
    assert (ref_run["fcalls"], ref_run["iters"]) == (g["fcalls"], g["iters"])
    assert ref_run["f"] == g["f"] and ref_run["x"] == g["x"]
    out = subprocess.check_output([os.path.join(built, "header_nm_lm"), mode, *_nm_args(g)],
                                  env=dict(env, NLSG_LIBRARY=LIB, NLSG_SUMMATION="tree"), text=True)
    o = json.loads(out)
    assert "device_error" not in o, o
    D = g["D"]
    x0 = hx(g["x0"]) + hx(g["x0_step"]) * np.arange(D, dtype=np.float64)
    kw = dict(step=hx(g["step"]), eps=hx(g["eps"]), max_iter=g["max_iter"], no_change=g["no_change"],
              restarts=g["restarts"], minimize=bool(g["minimize"]), order=1)
    if g["bounded"]:
        kw.update(upper=hx(g["upper"]), lower=hx(g["lower"]))
    ref, xr, _, _ = O.nm_run(oracle, x0, **kw)
    assert (o["fcalls"], o["iters"]) == (ref.function_calls_used, ref.iteration)
    assert hx(o["f"]) == ref.f_value
    assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,B,iters", [(96, 12, 3, 15), (120, 80, 2, 6)])
def test_lm_device_model_through_header_matches_oracle(built, oracle, m, n, B, iters):
    import math
    out = subprocess.check_output([os.path.join(built, "header_nm_lm"), "lm-device", str(m), str(n),
                                   str(B), str(iters)], env=dict(os.environ, NLSG_LIBRARY=LIB),
                                  text=True)
    res = json.loads(out)
    assert isinstance(res, list) and len(res) == B, res
    for p, o in enumerate(res):
        star = np.array([math.sin(0.37 * (j + 3 * p) + 0.1) for j in range(n)])
        A = np.array([[math.cos(0.11 * (i * n + j) + 1.3 * p) / math.sqrt(n) for j in range(n)]
                      for i in range(m)])
        y = np.zeros(m)
        for i in range(m):
            z = 0.0
            for j in range(n):
                z += A[i, j] * star[j]
            y[i] = math.tanh(z)
        th0 = np.array([0.5 * star[j] + 0.05 * math.cos(float(j)) for j in range(n)])
        ref, xr, lam, _ = O.lm_solve(oracle, A, y, th0, max_iter=iters, f_delta=0.0, order=1)
        assert (o["iters"], o["fcalls"], o["gcalls"], o["hcalls"]) == \
            (ref.iteration, ref.function_calls_used, ref.gradient_evals_used, ref.hessian_evals_used)
        assert hx(o["f"]) == ref.f_value
        assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["lm-device-fd", "lm-device-fd-custom"])
@pytest.mark.parametrize("n,iters", [(2, 12), (5, 8), (16, 4), (70, 2)])
def test_lm_default_functors_on_device_objective_through_header(built, oracle, n, iters, mode):
    """LevenbergMarquardt<device::Rosenbrock<double>, double>(f).minimize(x): fin_diff and
    fin_diff_h (nlsolver.h:3494-3511) evaluated on the GPU; with NLSG_SUMMATION=tree (a Custom objective
    always) bit-exact vs the tree oracle."""
    out = subprocess.check_output(
        [os.path.join(built, "header_nm_lm"), mode, str(n), "10", str(iters), "0.0",
         "0.8", "0.01"], env=dict(os.environ, NLSG_LIBRARY=LIB, NLSG_SUMMATION="tree"), text=True)
    o = json.loads(out)
    assert "device_error" not in o, o
    x0 = 0.8 + 0.01 * np.arange(n, dtype=np.float64)
    ref, xr, _, _ = O.lm_fd(oracle, "rosenbrock", x0, lam=10.0, max_iter=iters, f_delta=0.0, order=1)
    assert (o["iters"], o["fcalls"], o["gcalls"], o["hcalls"]) == \
        (ref.iteration, ref.function_calls_used, ref.gradient_evals_used, ref.hessian_evals_used)
    assert hx(o["f"]) == ref.f_value
    assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rosenbrock_n2_example_start", "rosenbrock_n4_near_minimum",
                                  "rosenbrock_n4_indefinite_nan", "rosenbrock_n16_6iters", "sphere_n5",
                                  "styblinski_tang_n8", "sphere_n64_3iters_lambda1", "rosenbrock_n100_2iters",
                                  "styblinski_tang_n130_2iters"])
def test_lm_minimize_through_header_is_the_reference_run(built, golden, name):
    """The drop-in's default: LevenbergMarquardt<device::Rosenbrock<double>, double>(f, ...).minimize(x)
    solves in reference order and returns the reference's run (tests/golden/lm_fd.json, made by the
    unmodified reference) bit for bit — the run that ends in NaN and the ones past 64 parameters
    included."""
    g = golden("lm_fd.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_nm_lm"), "lm-device-fd", str(g["n"]), repr(hx(g["lambda"])),
         str(g["max_iter"]), repr(hx(g["f_delta"])), repr(hx(g["x0"])), repr(hx(g["x0_step"])),
         OBJECTIVE_NAMES[g["objective"]]], env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in o, o
    assert (o["iters"], o["fcalls"], o["gcalls"], o["hcalls"]) == \
        (g["iters"], g["fcalls"], g["gcalls"], g["hcalls"])
    if np.isnan(hx(g["f"])):
        assert np.isnan(hx(o["f"])) and all(np.isnan(hx(v)) for v in o["x"])
    else:
        assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rosenbrock_n2_example_start", "rosenbrock_n4_near_minimum",
                                  "rosenbrock_n16_6iters", "rosenbrock_n100_2iters"])
def test_lm_custom_terms_through_header_run_in_index_order(built, golden, name):
    """The Rosenbrock chain as source text through LevenbergMarquardt's default functors: summed in index
    order by default, hence the reference's run bit for bit (one wave per problem, and past 64 parameters)."""
    g = golden("lm_fd.json")[name]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_nm_lm"), "lm-device-fd-custom", str(g["n"]), repr(hx(g["lambda"])),
         str(g["max_iter"]), repr(hx(g["f_delta"])), repr(hx(g["x0"])), repr(hx(g["x0_step"]))],
        env=dict(env, NLSG_LIBRARY=LIB), text=True))
    assert "device_error" not in o, o
    assert (o["iters"], o["fcalls"], o["gcalls"], o["hcalls"]) == \
        (g["iters"], g["fcalls"], g["gcalls"], g["hcalls"])
    assert o["f"] == g["f"] and o["x"] == g["x"]


@pytest.mark.gpu
def test_lm_summation_switch_through_header(built, oracle, golden):
    g = golden("lm_fd.json")["rosenbrock_n16_6iters"]
    cmd = [os.path.join(built, "header_nm_lm"), "lm-device-fd", str(g["n"]), repr(hx(g["lambda"])),
           str(g["max_iter"]), repr(hx(g["f_delta"])), repr(hx(g["x0"])), repr(hx(g["x0_step"])), "rosenbrock"]
    env = {k: v for k, v in os.environ.items() if k != "NLSG_SUMMATION"}
    env["NLSG_LIBRARY"] = LIB
    x0 = hx(g["x0"]) + hx(g["x0_step"]) * np.arange(g["n"], dtype=np.float64)
    tree, xt, _, _ = O.lm_fd(oracle, "rosenbrock", x0, lam=hx(g["lambda"]), max_iter=g["max_iter"],
                             f_delta=hx(g["f_delta"]), order=1)
    run = lambda extra, **kw: json.loads(subprocess.check_output(cmd + extra, env=dict(env, **kw), text=True))
    is_tree = lambda o: hx(o["f"]) == tree.f_value and np.array_equal(np.array([hx(v) for v in o["x"]]), xt)
    is_reference = lambda o: o["f"] == g["f"] and o["x"] == g["x"]
    assert tree.f_value != hx(g["f"])
    assert is_reference(run(["batch"]))  # (n <= 64: reference order is the faster evaluation, batches too)
    assert is_reference(run(["batch"], NLSG_SUMMATION="reference"))
    assert is_tree(run(["batch"], NLSG_SUMMATION="tree"))
    assert is_tree(run([], NLSG_SUMMATION="tree"))
    assert is_reference(run([]))
    # past 64 parameters (a workgroup per problem): the same rule
    gw = golden("lm_fd.json")["rosenbrock_n100_2iters"]
    cw = [os.path.join(built, "header_nm_lm"), "lm-device-fd", str(gw["n"]), repr(hx(gw["lambda"])),
          str(gw["max_iter"]), repr(hx(gw["f_delta"])), repr(hx(gw["x0"])), repr(hx(gw["x0_step"])), "rosenbrock"]
    one = json.loads(subprocess.check_output(cw, env=env, text=True))
    many = json.loads(subprocess.check_output(cw + ["batch"], env=env, text=True))
    assert one["f"] == gw["f"] and one["x"] == gw["x"]
    assert many["f"] == gw["f"] and many["x"] == gw["x"]
    tree = json.loads(subprocess.check_output(cw + ["batch"], env=dict(env, NLSG_SUMMATION="tree"), text=True))
    assert tree["f"] != gw["f"] and abs(hx(tree["f"]) - hx(gw["f"])) <= 1e-6 * abs(hx(gw["f"]))


def test_lm_device_objective_without_library_fails_loudly(built):
    r = subprocess.run([os.path.join(built, "header_nm_lm"), "lm-device-fd", "4", "10", "5", "0.0",
                        "0.8", "0.01"], env=dict(os.environ, NLSG_LIBRARY="/nonexistent/lib.so"),
                       capture_output=True, text=True)
    assert r.returncode == 3 and "device_error" in r.stdout, (r.returncode, r.stdout, r.stderr)


SANN_CASES = ["rosenbrock_n2_default_schedule", "rosenbrock_n8", "sphere_n16_hot",
              "styblinski_tang_n6", "sphere_n4_maximize", "rosenbrock_n130_ragged",
              "temp_iter_1_no_moves"]


@pytest.mark.parametrize("name", SANN_CASES)
def test_sann_host_path_through_header_matches_reference_bit_exact(built, golden, name):
    g = golden("sann.json")[name]
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_sann"), "host", str(g["objective"]), str(g["n"]),
         str(g["max_iter"]), str(g["temp_iter"]), repr(hx(g["temp_max"])), repr(hx(g["x0"])),
         repr(hx(g["x0_step"])), str(g["minimize"])], text=True))
    assert (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"])
    assert o["f"] == g["f"] and o["x"] == g["x"] and o["next_draw"] == g["next_draw"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "device-custom"])
@pytest.mark.parametrize("n,max_iter,temp_iter,minimize", [(2, 150, 10, 1), (16, 40, 6, 1),
                                                           (130, 12, 10, 0)])
def test_sann_device_objective_through_header_matches_oracle(built, oracle, mode, n, max_iter,
                                                             temp_iter, minimize):
    """SANN<device::Rosenbrock<double>, xorshift<double>>(f, gen, ...).minimize_batch(xs): chains
    keyed by (two draws of the generator, chain index); bit-exact vs the synchronous oracle."""
    B = 3
    out = subprocess.check_output(
        [os.path.join(built, "header_sann"), mode, str(B), str(n), str(max_iter), str(temp_iter),
         "10.0", "0.4", "0.01", str(minimize)], env=dict(os.environ, NLSG_LIBRARY=LIB), text=True)
    res = json.loads(out)
    assert isinstance(res, list) and len(res) == B, res
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    half = [min(int(oracle.orc_xorshift_next(C.byref(xs)) * 2.0**32), 2**32 - 1) for _ in range(2)]
    seed = (half[0] << 32) | half[1]
    for b, o in enumerate(res):
        x0 = 0.4 + 0.01 * (np.arange(n, dtype=np.float64) + b)
        ref, xr, _ = O.sann_sync(oracle, "rosenbrock", x0, seed, b, minimize=bool(minimize),
                                 max_iter=max_iter, temp_iter=temp_iter, temp_max=10.0)
        assert (o["fcalls"], o["iters"]) == (ref.function_calls_used, ref.iteration)
        assert hx(o["f"]) == ref.f_value
        assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


def test_sann_device_objective_without_library_fails_loudly(built):
    r = subprocess.run([os.path.join(built, "header_sann"), "device", "2", "4", "10", "10", "10.0",
                        "0.4", "0.01", "1"], env=dict(os.environ, NLSG_LIBRARY="/nonexistent/lib.so"),
                       capture_output=True, text=True)
    assert r.returncode == 3 and "device_error" in r.stdout, (r.returncode, r.stdout, r.stderr)


HYBRID_CASES = ["rosenbrock_n2_defaults", "rosenbrock_n4", "rosenbrock_n8_200iters", "rosenbrock_n16",
                "sphere_n6", "styblinski_tang_n4_maximize", "rosenbrock_n130_ragged"]


@pytest.mark.parametrize("name", HYBRID_CASES)
def test_hybrid_host_path_through_header_matches_reference_bit_exact(built, golden, name):
    g = golden("nmpso.json")[name]
    o = json.loads(subprocess.check_output(
        [os.path.join(built, "header_hybrid"), "host", str(g["objective"]), str(g["n"]),
         str(g["max_iter"]), repr(hx(g["eps"])), str(g["no_change"]), repr(hx(g["x0"])),
         repr(hx(g["x0_step"])), str(g["minimize"])], text=True))
    assert (o["fcalls"], o["iters"]) == (g["fcalls"], g["iters"])
    assert o["f"] == g["f"] and o["x"] == g["x"] and o["next_draw"] == g["next_draw"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "device-custom"])
@pytest.mark.parametrize("n,max_iter,eps,no_change", [(2, 1000, 1e-6, 20), (7, 60, 0.0, 1000),
                                                      (32, 20, 0.0, 1000)])
def test_hybrid_device_objective_through_header_matches_oracle(built, oracle, mode, n, max_iter, eps,
                                                               no_change):
    """NelderMeadPSO<device::Rosenbrock<double>, xorshift<double>>(f, gen, ...).minimize_batch(xs):
    instances keyed by (two draws of the generator, instance index); bit-exact vs the oracle."""
    B = 3
    out = subprocess.check_output(
        [os.path.join(built, "header_hybrid"), mode, str(B), str(n), str(max_iter), repr(eps),
         str(no_change), "0.4", "0.03"], env=dict(os.environ, NLSG_LIBRARY=LIB), text=True)
    res = json.loads(out)
    assert isinstance(res, list) and len(res) == B, res
    xs = O.XorShift()
    oracle.orc_xorshift_init(C.byref(xs))
    half = [min(int(oracle.orc_xorshift_next(C.byref(xs)) * 2.0**32), 2**32 - 1) for _ in range(2)]
    seed = (half[0] << 32) | half[1]
    for b, o in enumerate(res):
        x0 = 0.4 + 0.03 * (np.arange(n, dtype=np.float64) + b)
        ref, xr, _ = O.nmpso_sync(oracle, "rosenbrock", x0, seed, b, eps=eps, max_iter=max_iter,
                                  no_change=no_change)
        assert (o["fcalls"], o["iters"]) == (ref.function_calls_used, ref.iteration)
        assert hx(o["f"]) == ref.f_value
        assert np.array_equal(np.array([hx(v) for v in o["x"]]), xr)


def test_hybrid_device_objective_without_library_fails_loudly(built):
    r = subprocess.run([os.path.join(built, "header_hybrid"), "device", "2", "4", "10", "0.0", "20",
                        "0.4", "0.01"], env=dict(os.environ, NLSG_LIBRARY="/nonexistent/lib.so"),
                       capture_output=True, text=True)
    assert r.returncode == 3 and "device_error" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _tinyqr_prog(built, mode, header, arrays):
    text = header + "\n" + "\n".join(" ".join(float(v).hex() for v in np.ravel(a)) for a in arrays) + "\n"
    r = subprocess.run([os.path.join(built, "header_tinyqr"), mode], input=text, capture_output=True,
                       text=True, env=dict(os.environ, NLSG_LIBRARY=LIB))
    out = {}
    for line in r.stdout.splitlines():
        name, *vals = line.split()
        out[name] = np.array([float.fromhex(v) for v in vals])
    return r, out


def _fnv(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a, dtype=np.float64).view(np.uint8).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_tinyqr_namespace_host_path_matches_reference_bit_exact(built, oracle, golden):
    """namespace tinyqr of the drop-in header (qr_decomposition, back_solve, lm with the reference's
    signatures and defaults, tinyqr.h:291-310, 437-470) on the reference's own outputs: the 4 x 2
    example of SURVEY §3.5, the rectangular systems of tests/golden/tinyqr.json, the 64 x 64 damped
    system of lm.json — Q, R and beta bit for bit."""
    from tests.test_oracle_lm_golden import RECT, rect_system
    g = golden("lm.json")["tinyqr_example"]
    r, out = _tinyqr_prog(built, "host", "4 2", [[1, 1, 1, 1, 0, 1, 2, 3], [1, 3, 5, 7.5]])
    assert r.returncode == 0, r.stderr
    assert out["Q"].tolist() == [hx(v) for v in g["Q"]] and out["R"].tolist() == [hx(v) for v in g["R"]]
    assert out["beta"].tolist() == [hx(v) for v in g["beta"]]
    for n, p, seed in RECT:
        gg = golden("tinyqr.json")[f"rect_{n}x{p}"]
        X, y = rect_system(oracle, n, p, seed)
        r, out = _tinyqr_prog(built, "host", f"{n} {p}", [X, y])
        assert r.returncode == 0, r.stderr
        assert out["beta"].tolist() == [hx(v) for v in gg["beta"]], (n, p)
        assert out["beta_tol1e-8"].tolist() == [hx(v) for v in gg["beta_tol1e-8"]], (n, p)
        assert _fnv(out["Q"]) == int(gg["Q_fnv"]) and _fnv(out["R"]) == int(gg["R_fnv"]), (n, p)
    # the damped square system of the LM path (lm.json linalg_n64: M = B^T B + lambda I)
    gl, n = golden("lm.json")["linalg_n64"], 64
    k = oracle.orc_ctr_key(3, 77)
    B = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, e)) - 1 for e in range(n * n)]).reshape(n, n)
    b = np.array([2 * oracle.orc_u01(oracle.orc_ctr_key(k, n * n + i)) - 1 for i in range(n)])
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            acc = 0.0
            for l in range(n):
                acc += B[l, i] * B[l, j]
            M[i, j] = acc + (10.0 if i == j else 0.0)
    r, out = _tinyqr_prog(built, "host", f"{n} {n}", [M.T.reshape(-1), b])
    assert out["beta"].tolist() == [hx(v) for v in gl["qr_solution"]]
    assert _fnv(out["Q"]) == int(gl["Q_fnv"]) and _fnv(out["R"]) == int(gl["R_fnv"])


def test_tinyqr_device_has_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, _ = _tinyqr_prog(built, "device", "1 4 2", [[1, 1, 1, 1, 0, 1, 2, 3], [1, 3, 5, 7.5]])
    assert r.returncode == 4 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_tinyqr_device_batch_through_header(built, oracle):
    """tinyqr::device::lm: a batch of systems through the header -> dlopen -> C-ABI; every system
    equals the order-1 oracle bit for bit."""
    rng = np.random.default_rng(11)
    batch, n, p = 6, 50, 9
    X = 2 * rng.random((batch, p, n)) - 1
    y = 2 * rng.random((batch, n)) - 1
    r, out = _tinyqr_prog(built, "device", f"{batch} {n} {p}", [X, y])
    assert r.returncode == 0, r.stderr
    beta = out["beta"].reshape(batch, p)
    for b in range(batch):
        ref = np.zeros(p)
        oracle.orc_tinyqr_lm_order(O._ptr(np.ascontiguousarray(X[b].reshape(-1))),
                                   O._ptr(np.ascontiguousarray(y[b])), n, p, O._ptr(ref), 1)
        assert np.array_equal(beta[b], ref), b
