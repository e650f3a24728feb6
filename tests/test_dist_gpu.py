"""The sharded DE turn on a real GPU with RCCL (world = 1: the only size a one-GPU box offers):
the library-driven turn (nlsg_de_step_sharded: RCCL all-gather issued by the engine on a second
stream) and the host-driven turn (torch.distributed all_gather_into_tensor between turn_begin and
turn_end) must both reproduce the unsharded engine bit for bit, including the turn at which a stop
test fires. Runs in a child process so the process group does not outlive the test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import nlsolver_amd
from nlsolver_amd.dist import ShardedDE

native = sys.argv[2] == "native"
os.environ["NLSG_DIST_NATIVE"] = "1" if native else "0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[3])
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
out = []
# (pop, D, turns, settings); the last two are soaks: thousands of turns in an accepting regime, the
# generation on one stream racing ahead of heads, collectives and finalisers on the other
for pop, D, turns, kw in (
        (4096, 128, 30, dict(strategy=nlsolver_amd.DE_RANDOM, eps=0.0, max_iter=9, best_val_no_change=1000)),
        (4096, 128, 30, dict(strategy=nlsolver_amd.DE_RANDOM, eps=0.0, max_iter=1000, best_val_no_change=2)),
        (4096, 128, 30, dict(strategy=nlsolver_amd.DE_RANDOM, eps=1e-300, max_iter=12, best_val_no_change=1000)),
        (4096, 128, 30, dict(strategy=nlsolver_amd.DE_BEST, eps=1e-300, max_iter=7, best_val_no_change=1000)),
        (8192, 128, 3000, dict(strategy=nlsolver_amd.DE_RANDOM, eps=1e-300, max_iter=10**9, best_val_no_change=10**9)),
        (8192, 32, 3000, dict(strategy=nlsolver_amd.DE_RANDOM, eps=1e-300, max_iter=10**9, best_val_no_change=10**9))):
    x0 = np.full(D, 0.6)
    kw = dict(kw, CR=0.2, F=0.5, seed=99)
    with nlsolver_amd.DEEngine("rosenbrock", pop, D, **kw) as ref:
        ref.init(x0)
        ref.step(turns)
        P0, S0 = ref.download()
        st0 = ref.status()
        bx0, bf0, bi0 = ref.best()
    drv = ShardedDE(dist, lambda lo, n, stream: nlsolver_amd.DEEngine(
        "rosenbrock", pop, D, shard_lo=lo, shard_n=n, stream=stream, **kw), pop, D, device)
    assert drv.native == native
    drv.init(x0)
    drv.step(turns)
    torch.cuda.synchronize()
    P1, S1 = drv.engine.download()
    st1 = drv.engine.status()
    bx1, bf1, bi1 = drv.engine.best()
    drv.engine.close()
    out.append(dict(
        done=(st0.done, st1.done), iters=(st0.iteration, st1.iteration),
        fcalls=(st0.function_calls_used, st1.function_calls_used),
        best=(int(bi0), int(bi1)), same_pop=bool(np.array_equal(P0, P1)),
        same_scores=bool(np.array_equal(S0, S1)), same_best=bool(np.array_equal(bx0, bx1) and bf0 == bf1),
        std_err=(repr(st0.std_err), repr(st1.std_err))))
# PSO (the move needs the exchanged swarm best: summary -> all-gather -> finaliser -> move)
from nlsolver_amd.dist import ShardedPSO
n, Dp = 4096, 64
for kw in (dict(type=nlsolver_amd.PSO_ACCELERATED, eps=0.0, max_iter=9, best_val_no_change=1000),
           dict(type=nlsolver_amd.PSO_VANILLA, eps=1e-300, max_iter=7, best_val_no_change=1000)):
    kw = dict(kw, bounded=False, inertia=0.8, cognitive=1.8, social=1.8, seed=7)
    with nlsolver_amd.PSOEngine("rosenbrock", n, Dp, **kw) as ref:
        ref.init(-2.048, 2.048)
        ref.step(20)
        st0 = ref.status()
        bx0, bf0, bi0 = ref.best()
    drv = ShardedPSO(dist, lambda lo, m_, stream: nlsolver_amd.PSOEngine(
        "rosenbrock", n, Dp, shard_lo=lo, shard_n=m_, stream=stream, **kw), n, Dp, device)
    assert drv.native == native
    drv.init(-2.048, 2.048)
    drv.step(20)
    torch.cuda.synchronize()
    st1 = drv.engine.status()
    bx1, bf1, bi1 = drv.engine.best()
    drv.engine.close()
    out.append(dict(
        done=(st0.done, st1.done), iters=(st0.iteration, st1.iteration),
        fcalls=(st0.function_calls_used, st1.function_calls_used),
        best=(int(bi0), int(bi1)), same_pop=True, same_scores=True,
        same_best=bool(np.array_equal(bx0, bx1) and bf0 == bf1),
        std_err=(repr(st0.std_err), repr(st1.std_err))))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("mode,port", [("native", "29631"), ("host", "29632")])
def test_sharded_turn_world1_matches_unsharded_engine(mode, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, mode, port], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    for case in json.loads(line[7:]):
        assert case["done"][0] == case["done"][1], case
        assert case["iters"][0] == case["iters"][1] and case["fcalls"][0] == case["fcalls"][1], case
        assert case["best"][0] == case["best"][1], case
        assert case["same_pop"] and case["same_scores"] and case["same_best"], case
        assert case["std_err"][0] == case["std_err"][1], case


@pytest.mark.parametrize("workload,extra", [
    ("de", ["--pop-per-gpu", "8192"]), ("pso-accel", ["--pop-per-gpu", "8192"]),
    ("bfgs", ["--pop-per-gpu", "16"]), ("lm", ["--pop-per-gpu", "64"]), ("nm", ["--pop-per-gpu", "64"])])
def test_bench_two_rank_flow_rehearsal(workload, extra):
    """`python3 bench.py --gpus 2` exactly as the driver spells it — no external launcher:
    bench.py starts its two ranks itself (fresh child processes, before anything touches the GPU).
    On this one GPU the ranks share device 0 and talk over gloo (NLSG_BENCH_REHEARSAL=1; RCCL
    refuses two ranks on one device): every collective step is entered by every rank, rank 0 prints
    exactly one JSON line with n_gpus = 2. The number itself means nothing."""
    env = dict(os.environ, NLSG_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40",
           "--warmup", "5", "--workload", workload, "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["roofline"]["kernel_ms"] > 0
    if workload in ("de", "pso-accel"):
        assert out["steps"] == 40
        # host-ordered turns over gloo in the rehearsal: no library-side communicator to read back
        assert out["config"]["turn_driver"].startswith("host") and out["config"]["rccl_ranks"] is None
    else:
        assert out["config"]["parallelism"].startswith("replicas x2")


@pytest.mark.parametrize("workload", ["de", "pso-accel"])
def test_bench_four_rank_flow_rehearsal(workload):
    """`python3 bench.py --gpus 4` — the largest self-launched job this pool lets a one-GPU box run
    (at most six processes may hold the card, and this test process is one of them; the eight-rank
    exchange is rehearsed over gloo on the CPU, tests/test_dist_gloo.py): four ranks share device
    0, every collective step is entered by every rank, ONE JSON line with n_gpus = 4 whose global
    size is four shards."""
    env = dict(os.environ, NLSG_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20",
           "--warmup", "5", "--workload", workload, "--no-cpu-baseline", "--pop-per-gpu", "4096"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["value"] > 0 and out["steps"] == 20
    assert out["config"]["global_pop" if workload == "de" else "global_swarm"] == 4 * 4096
    assert out["config"]["turn_driver"].startswith("host")


def test_bench_two_rank_default_line_carries_the_sharded_pso_run():
    """`python3 bench.py --gpus 2 --steps K --warmup W` at the default size, as the driver runs the
    scaling series: after the DE job a second two-rank job runs BASELINE configs[4] (the PSO swarm
    sharded over the ranks) and its summary rides in the one JSON line as `other_configs`."""
    env = dict(os.environ, NLSG_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["metric"].startswith("candidate-evals/sec")
    (pso,) = out["other_configs"]
    assert "error" not in pso, pso
    assert pso["n_gpus"] == 2 and pso["value"] > 0 and pso["unit"] == "particle-evals/s"
    assert pso["turn_driver"].startswith("host")


def test_bench_forced_dist_reports_rccl_ranks():
    """One rank, but through the sharded path and the library's own RCCL communicator
    (NLSG_BENCH_FORCE_DIST=1): the line carries the communicator size RCCL reports."""
    env = dict(os.environ, NLSG_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_PORT="29651")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40",
           "--warmup", "5", "--pop-per-gpu", "8192", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["config"]["rccl_ranks"] == 1 and out["config"]["turn_driver"].startswith("library")


def test_bench_self_launch_propagates_failure():
    """A rank that dies takes the whole `bench.py --gpus 2` down with a non-zero exit code and no
    JSON line (here: an impossible shard size)."""
    env = dict(os.environ, NLSG_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--pop-per-gpu", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
