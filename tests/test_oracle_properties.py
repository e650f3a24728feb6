"""CPU checks of the synchronous restatement itself (no GPU):
tree objective vs reference arithmetic (1e-12), reduction trees, convergence
of the synchronous algorithm to the reference's results on config C1."""
import ctypes as C

import numpy as np
import pytest

from tests import _oracle as O


@pytest.mark.parametrize("obj", ["rosenbrock", "sphere", "styblinski_tang", "rastrigin"])
@pytest.mark.parametrize("D", [1, 2, 3, 5, 64, 127, 128, 129, 256, 1000, 1024])
def test_tree_objective_within_1e12_of_sequential(oracle, obj, D):
    rng = np.random.default_rng(D)
    for _ in range(20):
        x = rng.uniform(-2.048, 2.048, D)
        seq = oracle.orc_objective_seq(O.OBJ[obj], x.ctypes.data_as(O.pd), D)
        tree = oracle.orc_objective_tree(O.OBJ[obj], x.ctypes.data_as(O.pd), D)
        assert abs(tree - seq) <= 1e-12 * max(1.0, abs(seq))  # north_star tolerance
    if D == 2:  # a single term: the tree is exact
        assert tree == seq


@pytest.mark.parametrize("n", [2, 3, 255, 256, 257, 1024, 1025, 4096, 65536 + 7])
def test_std_err_tree_close_to_reference_formula(oracle, n):
    x = np.random.default_rng(n).normal(50, 10, n)
    ser = oracle.orc_std_err_serial(x.ctypes.data_as(O.pd), n)
    tree = oracle.orc_std_err_tree(x.ctypes.data_as(O.pd), n)
    assert abs(ser - tree) <= 1e-12 * ser
    assert abs(ser - np.std(x, ddof=1)) <= 1e-12 * ser


def test_sync_de_reaches_reference_quality_on_c1(oracle, golden):
    """Config C1 (Rosenbrock-2D, pop=40, x0={5,7}): the synchronous counter-RNG
    algorithm lands where the reference lands (its own tests accept 0.05)."""
    ref = golden("de_c1.json")["c1_random_pop40_x0_5_7"]
    run = O.DESyncRun(oracle, "rosenbrock", 40, 2, [5, 7], eps=10e-4)
    for _ in range(2000):
        run.step()
        if run.s.done:
            break
    assert run.s.done
    x = run.best_x
    assert np.all(np.abs(x - 1.0) <= 0.05)
    assert run.scores[run.s.best_id] < 1e-3
    # same order of magnitude of work as the reference (1840 calls / 45 iterations)
    assert 0.3 * ref["iters"] <= run.s.iter <= 3 * ref["iters"]
    assert run.s.fcalls == 40 * (run.s.iter + 1)


def test_sync_de_island_shards_keep_donors_local(oracle):
    run = O.DESyncRun(oracle, "rosenbrock", 64, 8, [4.096] * 8, n_shards=4, trace=True)
    run.step()
    tr = run.trace
    for a in range(64):
        lo = (a // 16) * 16
        r = tr[a, :3]
        assert np.all((r >= lo) & (r < lo + 16)) and len(set(r.tolist()) | {a}) == 4


def test_omp_variant_is_bit_identical(oracle):
    a = O.DESyncRun(oracle, "rosenbrock", 512, 128, [4.096] * 128)
    b = O.DESyncRun(oracle, "rosenbrock", 512, 128, [4.096] * 128)
    a.step(3)
    b.step(3, threads=4)
    assert np.array_equal(a.population, b.population) and np.array_equal(a.scores, b.scores)


@pytest.mark.parametrize("strategy", [1, 0])
@pytest.mark.parametrize("pop,D", [(64, 16), (256, 128), (37, 5)])
def test_sync_donors_and_trials_are_the_reference_logic_on_the_same_draws(oracle, strategy, pop, D):
    """The link between the two parity chains (device == synchronous restatement bit for bit;
    serial restatement == the reference bit for bit): the synchronous generation's donors, forced
    dimension and trial vector are what the REFERENCE'S OWN code path — generate_indices
    (nlsolver.h:2331-2355) and propose_new_agent (:2357-2375) as orc_de_serial runs them, pinned to
    the goldens — makes of the same uniform draws. The keyed draws of agent a (slots D + 1 + k:
    donor proposals, D: forced dimension, 0 .. D - 1: crossover) are replayed into that code."""
    import ctypes as C
    x0 = np.full(D, 0.6) * (1.0 + 0.001 * np.arange(D))
    run = O.DESyncRun(oracle, "rosenbrock", pop, D, x0, strategy=strategy, CR=0.2, F=0.5, eps=0.0,
                      max_iter=1000, best_val_no_change=1000, trace=True)
    run.step(2)                               # generation 2 has seen acceptances
    before = run.population.copy()
    scores_before = run.scores.copy()
    gen = run.s.iter + 1                      # the generation about to run
    run.step(1)
    after, trace = run.population, run.trace
    best_of_turn = run.s.best_id              # found by the turn's scan, before its generation (:2432-2437)
    kg = oracle.orc_ctr_key(run.s.seed, gen)
    sz = C.c_size_t
    fn = oracle.orc_de_serial_proposal_from_draws
    fn.restype = sz
    fn.argtypes = [O.pd, sz, sz, sz, C.c_double, C.c_double, O.pd, sz, O.pd, C.POINTER(sz),
                   C.POINTER(sz), O.pd]
    accepted = 0
    for a in range(pop):
        ka = oracle.orc_ctr_key(kg, a)
        donor = np.array([oracle.orc_u01(oracle.orc_ctr_key(ka, D + 1 + k)) for k in range(64)])
        cross = np.array([oracle.orc_u01(oracle.orc_ctr_key(ka, D))] +
                         [oracle.orc_u01(oracle.orc_ctr_key(ka, d)) for d in range(D)])
        ids, used, prop = (sz * 4)(), sz(), np.zeros(D)
        fixed = a if strategy == 1 else best_of_turn
        dim = fn(O._ptr(before), pop, D, fixed, 0.2, 0.5, O._ptr(donor), donor.size, O._ptr(cross), ids,
                 C.byref(used), O._ptr(prop))
        assert used.value <= 64
        assert [ids[1], ids[2], ids[3]] == trace[a, :3].tolist(), a
        assert dim == trace[a, 3], a
        if trace[a, 4]:                       # accepted: the survivor is the trial vector
            accepted += 1
            assert np.array_equal(after[a], prop), a
            assert run.scores[a] < scores_before[a]
        else:
            assert np.array_equal(after[a], before[a]), a
    assert accepted > 0


@pytest.mark.parametrize("bounded", [False, True])
def test_sync_pso_move_is_the_reference_update_on_the_same_normals(oracle, bounded):
    """Accelerated PSO: a particle's new position in the synchronous restatement (what the GPU
    executes) is the reference's update_positions + threshold_positions (nlsolver.h:2687-2715) — the
    code the serial restatement, pinned to the reference's runs, executes — applied to the same
    normal variates; the two restatements differ only in where the variates come from (one keyed
    64-bit draw each instead of two xorshift draws)."""
    import ctypes as C
    n, D = 48, 24
    kw = dict(type=O.PSO_ACCELERATED, bounded=bounded, eps=0.0, max_iter=1000, best_val_no_change=1000)
    run = O.PSOSyncRun(oracle, "rosenbrock", n, D, -1.5, 2.0, **kw)
    run.step(3)
    before = run.pos.copy()
    it = run.s.iter
    run.step(1)
    gbest = run.gbest_x.copy()  # as the move of this turn saw it (the head ran first)
    kg = oracle.orc_ctr_key(run.s.seed, it + 1)
    fn = oracle.orc_pso_accel_move_from_normals
    fn.restype = None
    fn.argtypes = [O.pd, O.pd, O.pd, O.pd, O.pd, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int]
    import math
    for i in range(n):
        kp = oracle.orc_ctr_key(kg, i)
        normals = np.array([oracle.orc_rnorm(oracle.orc_ctr_key(kp, 2 * j)) for j in range(D)])
        p = before[i].copy()
        fn(O._ptr(p), O._ptr(normals), O._ptr(gbest), O._ptr(run.lower), O._ptr(run.upper), D,
           math.pow(0.8, it), 1.8, 1.8, int(bounded))
        assert np.array_equal(p, run.pos[i]), i  # inertia = pow(init_inertia, iter), :2613
