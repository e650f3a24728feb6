"""Parity tests: the HIP PSO path (through the C-ABI) vs the oracle's synchronous
restatement. Bit-exact positions, velocities, personal bests, swarm best, counters."""
import numpy as np
import pytest

from tests import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod():
    import torch
    assert torch.cuda.is_available()
    import nlsolver_amd
    return nlsolver_amd


def check_state(eng, ref, tag):
    pos, vel, pbest, cur = eng.download()
    assert np.array_equal(pos, ref.pos), f"{tag}: positions"
    assert np.array_equal(cur, ref.cur_val), f"{tag}: values of the last evaluation"
    assert np.array_equal(pbest, ref.pbest_val), f"{tag}: personal-best values"
    if vel is not None:
        assert np.array_equal(vel, ref.vel), f"{tag}: velocities"
    st = eng.status()
    assert (st.iteration, st.function_calls_used, st.val_no_change, st.done) == \
        (ref.s.iter, ref.s.fevals, ref.s.val_no_change, ref.s.done), tag
    if ref.s.fevals:
        assert st.f_value == ref.s.gbest_val and st.best_index == ref.s.gbest_idx, tag


@pytest.mark.parametrize("n,D", [(10, 2), (64, 256), (37, 130), (16, 5), (1030, 128), (8, 1024)])
@pytest.mark.parametrize("type_", [O.PSO_ACCELERATED, O.PSO_VANILLA])
@pytest.mark.parametrize("bounded", [False, True])
def test_pso_turns_bit_exact(mod, oracle, n, D, type_, bounded):
    lo = -2.048 * (1 + 0.001 * np.arange(D))
    hi = 2.048 * (1 + 0.002 * np.arange(D))
    kw = dict(type=type_, bounded=bounded, eps=0.0, max_iter=1000, best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, lo, hi, **kw)
    with mod.PSOEngine("rosenbrock", n, D, **kw) as eng:
        eng.init(lo, hi)
        check_state(eng, ref, "init")
        for t in range(5):
            eng.step(1)
            ref.step(1)
            check_state(eng, ref, f"turn {t}")
        bx, bf, bi = eng.best()
        assert np.array_equal(bx, ref.gbest_x) and bf == ref.s.gbest_val and bi == ref.s.gbest_idx


@pytest.mark.parametrize("n,D", [(24, 1025), (20, 1026), (16, 2048), (12, 3001)])
@pytest.mark.parametrize("type_", [O.PSO_ACCELERATED, O.PSO_VANILLA])
@pytest.mark.parametrize("bounded", [False, True])
def test_pso_particles_longer_than_1024_coordinates_bit_exact(mod, oracle, n, D, type_, bounded):
    """D > 1024 (the reference has no limit, nlsolver.h:2498-2742): particles streamed in segments
    of 1024 coordinates, in place; first size past the old cap, an odd one, whole segments, a ragged
    last segment."""
    lo = -2.048 * (1 + 0.0001 * np.arange(D))
    hi = 2.048 * (1 + 0.0002 * np.arange(D))
    kw = dict(type=type_, bounded=bounded, eps=0.0, max_iter=1000, best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, lo, hi, **kw)
    with mod.PSOEngine("rosenbrock", n, D, **kw) as eng:
        eng.init(lo, hi)
        check_state(eng, ref, "init")
        for t in range(4):
            eng.step(1)
            ref.step(1)
            check_state(eng, ref, f"turn {t}")
        bx, bf, bi = eng.best()
        assert np.array_equal(bx, ref.gbest_x) and bf == ref.s.gbest_val and bi == ref.s.gbest_idx


@pytest.mark.parametrize("kw", [dict(eps=10e-4), dict(eps=0.0, max_iter=9),
                                dict(eps=0.0, best_val_no_change=3), dict(eps=50.0)])
@pytest.mark.parametrize("type_", [O.PSO_ACCELERATED, O.PSO_VANILLA])
def test_pso_full_minimize_matches_oracle_to_the_stop(mod, oracle, kw, type_):
    args = dict(eps=10e-4, max_iter=300, best_val_no_change=50, type=type_)
    args.update(kw)
    ref = O.PSOSyncRun(oracle, "rosenbrock", 10, 2, -3.0, 3.0, **args)
    while not ref.s.done:
        ref.step()
    x = np.zeros(2)
    with mod.PSOEngine("rosenbrock", 10, 2, **args) as eng:
        st = eng.minimize(x, -3.0, 3.0, poll_every=7)
    assert st.done == 1
    assert (st.iteration, st.function_calls_used) == (ref.s.iter, ref.s.fevals)
    assert st.f_value == ref.s.gbest_val and np.array_equal(x, ref.gbest_x)
    if args["eps"] > 0:
        assert st.std_err == ref.s.std_err


def test_pso_maximize_and_other_objectives(mod, oracle):
    for obj, mini in (("sphere", False), ("styblinski_tang", True)):
        kw = dict(type=O.PSO_ACCELERATED, minimize=mini, eps=0.0, max_iter=100, best_val_no_change=1000)
        ref = O.PSOSyncRun(oracle, obj, 48, 20, -2.0, 3.0, **kw)
        ref.step(6)
        with mod.PSOEngine(obj, 48, 20, **kw) as eng:
            eng.init(-2.0, 3.0)
            eng.step(6)
            check_state(eng, ref, obj)


def test_pso_class_mirror_reaches_reference_quality(mod, golden):
    """PSO(f, gen, ...) with the reference's defaults through the class mirror: Accelerated
    PSO on Rosenbrock-2D from x0=(3,3) lands near the reference's result quality."""
    ref = golden("pso.json")["accel_2d_x0_3_3"]
    x = np.array([3.0, 3.0])
    st = mod.PSO("rosenbrock", None, 0.8, 1.8, 1.8, 10, 50, 1000, 0.0,
                 type=mod.PSO_ACCELERATED).minimize(x)
    fcalls, iters, f, g, h = st.get_summary()
    assert (fcalls, iters) == (ref["fcalls"], ref["iters"]) == (510, 50)
    assert f < 0.5 and np.all(np.abs(x - 1.0) < 0.8)


@pytest.mark.parametrize("type_", [O.PSO_ACCELERATED, O.PSO_VANILLA])
@pytest.mark.parametrize("eps", [0.0, 200.0])
@pytest.mark.parametrize("D", [256, 24])  # one particle per wave / several per wave
def test_pso_sharded_path_on_one_gpu_bit_exact(mod, oracle, type_, eps, D):
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    n, shards, turns = 4096, 4, 6
    kw = dict(type=type_, eps=eps, max_iter=1000, best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, -2.048, 2.048, n_shards=shards, **kw)
    ref.step(turns)
    m = n // shards
    engs = [mod.PSOEngine("rosenbrock", n, D, shard_lo=r * m, shard_n=m, stream=stream, **kw)
            for r in range(shards)]
    rec = engs[0].record_doubles()
    gathered = torch.zeros(shards * rec, dtype=torch.float64, device=dev)
    for e in engs:
        e.init(-2.048, 2.048)
    for _ in range(turns):
        for r, e in enumerate(engs):
            e.turn_begin(gathered[r * rec:(r + 1) * rec].data_ptr())
        for e in engs:
            e.turn_end(gathered.data_ptr(), shards)
    for r, e in enumerate(engs):
        pos, vel, pbest, cur = e.download()
        sl = slice(r * m, (r + 1) * m)
        assert np.array_equal(pos, ref.pos[sl]) and np.array_equal(pbest, ref.pbest_val[sl])
        st = e.status()
        assert (st.iteration, st.function_calls_used, st.val_no_change, st.done, st.best_index) == \
            (ref.s.iter, ref.s.fevals, ref.s.val_no_change, ref.s.done, ref.s.gbest_idx)
        assert st.f_value == ref.s.gbest_val
        if eps > 0:
            assert st.std_err == ref.s.std_err
        bx, _, _ = e.best()
        assert np.array_equal(bx, ref.gbest_x)
        e.close()


def test_pso_config5_shard_size_properties(mod, oracle):
    """BASELINE config 5's per-GPU shard: 131072 particles x D=256 (2^20 / 8 GPUs)."""
    n, D = 131072, 256
    kw = dict(type=O.PSO_ACCELERATED, eps=0.0, max_iter=1000, best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, -2.048, 2.048, **kw)
    ref.step(2, threads=8)
    with mod.PSOEngine("rosenbrock", n, D, **kw) as eng:
        eng.init(-2.048, 2.048)
        eng.step(2)
        check_state(eng, ref, "config5 shard")


@pytest.mark.parametrize("type_", [O.PSO_ACCELERATED, O.PSO_VANILLA])
def test_pso_config5_global_size_eight_shards_on_one_gpu(mod, oracle, type_):
    """BASELINE configs[4] AS WORDED: swarm = 2^20 particles x D = 256 sharded 8 x 131072, the best
    record exchanged every iteration — all eight shard engines on the one device (2 GiB of
    positions), the records concatenated where RCCL's all-gather would put them. Exercises what no
    smaller rehearsal does: global particle ids >= 2^17 (the counter RNG is keyed by them),
    shard_lo >= 2^17 and the finaliser over eight records. Bit-compared with the restatement run
    with n_shards = 8 on the whole swarm (nlsolver.h:2593-2741 through SURVEY §8e's partitioning)."""
    import torch
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    n, D, shards, turns = 1 << 20, 256, 8, 2
    m = n // shards
    kw = dict(type=type_, eps=0.0, max_iter=1000, best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, "rosenbrock", n, D, -2.048, 2.048, n_shards=shards, **kw)
    ref.step(turns, threads=16)
    engs = [mod.PSOEngine("rosenbrock", n, D, shard_lo=r * m, shard_n=m, stream=stream, **kw)
            for r in range(shards)]
    rec = engs[0].record_doubles()
    gathered = torch.zeros(shards * rec, dtype=torch.float64, device=dev)
    for e in engs:
        e.init(-2.048, 2.048)
    for _ in range(turns):
        for r, e in enumerate(engs):
            e.turn_begin(gathered[r * rec:(r + 1) * rec].data_ptr())
        for e in engs:
            e.turn_end(gathered.data_ptr(), shards)
    records = gathered.cpu().numpy().reshape(shards, rec)
    # every rank sees the same eight records and draws the same conclusion from them
    stats, mins = [], []
    for r, e in enumerate(engs):
        pos, vel, pbest, cur = e.download()
        sl = slice(r * m, (r + 1) * m)
        assert np.array_equal(pos, ref.pos[sl]), f"shard {r}: positions"
        assert np.array_equal(pbest, ref.pbest_val[sl]), f"shard {r}: personal bests"
        if type_ == O.PSO_VANILLA:
            assert np.array_equal(vel, ref.vel[sl]), f"shard {r}: velocities"
        mins.append((pbest.min(), r * m + int(pbest.argmin())))
        st = e.status()
        stats.append((st.iteration, st.function_calls_used, st.val_no_change, st.done, st.best_index,
                      st.f_value))
        bx, _, _ = e.best()
        assert np.array_equal(bx, ref.gbest_x), f"shard {r}: swarm best"
        e.close()
    assert all(s == stats[0] for s in stats)
    assert stats[0] == (ref.s.iter, ref.s.fevals, ref.s.val_no_change, ref.s.done, ref.s.gbest_idx,
                        ref.s.gbest_val)
    assert stats[0][1] == n * turns and 0 <= stats[0][4] < n  # one evaluation per particle and turn
    # independent of the oracle: the swarm best every shard reports is the minimum over all
    # downloaded personal bests (a particle's personal best is the least value it ever had)
    assert np.isfinite(records).all()
    assert stats[0][5] == min(v for v, _ in mins)


def _sweep_cases(n=24, seed=20261004):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([1, 3, 4, 5, 63, 255, 256, 257, 1023, 1025, 2050])),
                    int(rng.choice([1, 2, 3, 31, 127, 128, 129, 256, 300])),
                    int(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                    float(rng.choice([0.0, 1e-300, 10e-4])), int(rng.integers(1, 2**31)),
                    str(rng.choice(["rosenbrock", "sphere", "styblinski_tang"]))))
    return out


@pytest.mark.parametrize("n,D,type_,bounded,minimize,eps,seed,obj", _sweep_cases())
def test_pso_randomised_configuration_sweep_bit_exact(mod, oracle, n, D, type_, bounded, minimize,
                                                      eps, seed, obj):
    """Swarm sizes around the block / tile boundaries, odd and multi-chunk dimensions, both types,
    bounded or not, minimise / maximise, std_err on and off: 6 turns equal the oracle's."""
    kw = dict(type=type_, bounded=bounded, minimize=minimize, eps=eps, seed=seed, max_iter=5,
              best_val_no_change=1000)
    ref = O.PSOSyncRun(oracle, obj, n, D, -1.5, 2.5, **kw)
    with mod.PSOEngine(obj, n, D, **kw) as eng:
        eng.init(-1.5, 2.5)
        for t in range(6):
            eng.step(1)
            ref.step(1)
        check_state(eng, ref, "after 6 turns")
        bx, bf, bi = eng.best()
        assert np.array_equal(bx, ref.gbest_x) and bf == ref.s.gbest_val and bi == ref.s.gbest_idx


@pytest.mark.parametrize("D,n", [(3, 37), (8, 1000), (16, 4096), (33, 515), (64, 2048)])
@pytest.mark.parametrize("type_", ["accelerated", "vanilla"])
def test_packing_does_not_change_the_history(mod, monkeypatch, D, n, type_):
    """Particles of at most 64 coordinates share a wave; NLSG_PSO_GROUPS=0 keeps one per wave."""
    t = mod.PSO_ACCELERATED if type_ == "accelerated" else mod.PSO_VANILLA
    out = []
    for groups in ("1", "0"):
        monkeypatch.setenv("NLSG_PSO_GROUPS", groups)
        with mod.PSOEngine("rosenbrock", n, D, type=t, bounded=True, eps=0.0, max_iter=10**9,
                           best_val_no_change=10**9, seed=99) as eng:
            eng.init(-1.5, 2.0)
            eng.step(10)
            st = eng.status()
            out.append((eng.download(), eng.best(), (st.iteration, st.f_value)))
    for u, v in zip(out[0][0], out[1][0]):
        assert (u is None and v is None) or np.array_equal(u, v)
    assert np.array_equal(out[0][1][0], out[1][1][0]) and out[0][1][1:] == out[1][1][1:]
    assert out[0][2] == out[1][2]
