"""N>1 path on CPU: world_size-2 and world_size-8 gloo processes drive nlsolver_amd.dist.ShardedDE /
ShardedPSO (the product's host-side sharding/exchange logic) over a CPU stand-in engine built on
the oracle. The result must equal the single-process restatement with n_shards = world. Eight
ranks — the scaling run's size, an 8-record finaliser on every rank — can only be rehearsed here:
a GPU box of this pool admits at most six processes on its card."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nlsolver_amd.dist import ShardedDE, shard_bounds
from tests import _oracle as O


def test_shard_bounds():
    assert [shard_bounds(64, 4, r) for r in range(4)] == [(0, 16), (16, 16), (32, 16), (48, 16)]
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cfg, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = O.load()
    pop, D, turns, kw = cfg["pop"], cfg["D"], cfg["turns"], cfg["kw"]
    drv = ShardedDE(dist, lambda lo, n, stream: O.OracleShardEngine(lib, "rosenbrock", pop, D, lo, n, **kw),
                    pop, D, torch.device("cpu"))
    drv.init(np.full(D, 4.096))
    drv.step(turns)
    P, S = drv.engine.shard()
    s = drv.engine.run.s
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=P, S=S,
             state=np.array([s.best_id, s.iter, s.val_no_change, s.fcalls, s.done], dtype=np.int64),
             std_err=np.array([s.std_err]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(strategy=1, eps=0.0, best_val_no_change=1000),
                                dict(strategy=0, eps=0.0, best_val_no_change=1000),
                                dict(strategy=1, eps=5000.0, best_val_no_change=1000),
                                dict(strategy=0, eps=0.0, best_val_no_change=2)])
@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_match_single_process_oracle(tmp_path, oracle, kw, world):
    pop, D, turns = 32 * world, 16, 12
    cfg = dict(pop=pop, D=D, turns=turns, kw=kw)
    mp.start_processes(_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world,
                       join=True, start_method="fork")
    ref = O.DESyncRun(oracle, "rosenbrock", pop, D, np.full(D, 4.096), n_shards=world, **kw)
    ref.step(turns)
    n = pop // world
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["P"], ref.population[r * n:(r + 1) * n]), f"rank {r} population"
        assert np.array_equal(got["S"], ref.scores[r * n:(r + 1) * n]), f"rank {r} scores"
        # every rank holds the same solver state after the same finaliser
        assert got["state"].tolist() == [ref.s.best_id, ref.s.iter, ref.s.val_no_change,
                                         ref.s.fcalls, ref.s.done]
        if kw["eps"] > 0:
            assert got["std_err"][0] == ref.s.std_err
    if kw.get("eps", 0) > 0 or kw.get("best_val_no_change", 1000) < 10:
        assert ref.s.done == 1 and ref.s.iter < turns  # the stop test really fired mid-run


def _pso_worker(rank, world, port, cfg, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nlsolver_amd.dist import ShardedPSO
    lib = O.load()
    n, D, turns, kw = cfg["n"], cfg["D"], cfg["turns"], cfg["kw"]
    drv = ShardedPSO(dist, lambda lo, m, stream: O.OraclePSOShardEngine(lib, "rosenbrock", n, D, lo, m, **kw),
                     n, D, torch.device("cpu"))
    drv.init(np.full(D, -2.048), np.full(D, 2.048))
    drv.step(turns)
    run = drv.engine.run
    sl = slice(drv.lo, drv.lo + drv.n)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=run.pos[sl], B=run.pbest_val[sl],
             G=run.gbest_x, state=np.array([run.s.gbest_idx, run.s.iter, run.s.val_no_change,
                                            run.s.fevals, run.s.done], dtype=np.int64),
             gval=np.array([run.s.gbest_val, run.s.std_err]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(type=O.PSO_ACCELERATED, eps=0.0, best_val_no_change=1000),
                                dict(type=O.PSO_VANILLA, eps=0.0, best_val_no_change=1000),
                                dict(type=O.PSO_ACCELERATED, eps=2000.0, best_val_no_change=1000)])
@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_pso_match_single_process_oracle(tmp_path, oracle, kw, world):
    n, D, turns = 32 * world, 16, 10
    cfg = dict(n=n, D=D, turns=turns, kw=kw)
    mp.start_processes(_pso_worker, args=(world, _free_port(), cfg, str(tmp_path)), nprocs=world,
                       join=True, start_method="fork")
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, -2.048, 2.048, n_shards=world, **kw)
    ref.step(turns)
    m = n // world
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["P"], ref.pos[r * m:(r + 1) * m])
        assert np.array_equal(got["B"], ref.pbest_val[r * m:(r + 1) * m])
        assert np.array_equal(got["G"], ref.gbest_x)
        assert got["state"].tolist() == [ref.s.gbest_idx, ref.s.iter, ref.s.val_no_change,
                                         ref.s.fevals, ref.s.done]
        assert got["gval"][0] == ref.s.gbest_val
        if kw["eps"] > 0:
            assert got["gval"][1] == ref.s.std_err


def test_bench_self_launch_without_gpu_fails_loudly():
    """`python bench.py --gpus 2` with no launcher starts its two ranks itself; without a GPU every
    rank refuses ("no CPU fallback") and the launcher must come back non-zero, with no JSON line,
    instead of hanging or printing a number."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU behaviour")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr and "needs a GPU" in r.stderr
    assert not r.stdout.strip()
