"""CPU leg of the distribution-level parity check (tests/_stat.py): the synchronous oracle — the
restatement the device engines reproduce bit for bit — sampled over K = 128 counter seeds against
the reference's own K = 128 seeded runs (tests/golden/de_stat.json, pso_stat.json).

The D = 128 configurations run on the device only (tests/test_stat_gpu.py): 1e8 objective
evaluations each is a minute of CPU per configuration.
"""
import numpy as np
import pytest

from tests import _oracle as O
from tests import _stat as S

DE = S.load("de_stat.json")
PSO = S.load("pso_stat.json")
CPU_DE = [n for n in DE["configs"] if "D128" not in n]
CPU_PSO = [n for n in PSO["configs"] if "D128" not in n]


def sample_de_oracle(lib, c, gens, K, threads=0):
    smp = S.Sample(gens)
    for k in range(K):
        r = O.DESyncRun(lib, "rosenbrock", c["pop"], c["D"], S.x0_of(c),
                        strategy=1 if c["strategy"] == "random" else 0, eps=c["eps"],
                        CR=c["CR"], F=c["F"], max_iter=c["max_iter"],
                        best_val_no_change=c["no_change"], seed=S.seed_of(k))
        S.run_marks(gens, c["max_iter"], lambda n: r.step(n, threads=threads),
                    lambda: (r.scores.min(), r.scores.mean()),
                    lambda: (int(r.s.iter), float(r.scores[r.s.best_id])), smp)
        assert r.s.done
    return smp.arrays()


def sample_pso_oracle(lib, c, gens, K, threads=1):
    smp = S.Sample(gens)
    x0 = S.x0_of(c)
    for k in range(K):
        r = O.PSOSyncRun(lib, "rosenbrock", c["particles"], c["D"], -np.abs(x0), np.abs(x0),
                         type=O.PSO_ACCELERATED, bounded=False, eps=c["eps"],
                         max_iter=c["max_iter"], best_val_no_change=c["no_change"],
                         seed=S.seed_of(k))
        S.run_marks(gens, c["max_iter"], lambda n: r.step(n, threads=threads),
                    lambda: (r.pbest_val.min(), r.pbest_val.mean()),
                    lambda: (int(r.s.iter), float(r.s.gbest_val)), smp)
        assert r.s.done
    return smp.arrays()


@pytest.mark.parametrize("name", CPU_DE)
def test_de_sync_oracle_vs_reference_distribution(name):
    c = DE["configs"][name]
    smp = sample_de_oracle(O.load(), c, DE["gens"], DE["K"], threads=4 if c["pop"] >= 1024 else 0)
    S.compare(name, c, smp, DE["gens"])


@pytest.mark.parametrize("name", CPU_PSO)
def test_pso_sync_oracle_vs_reference_distribution(name):
    c = PSO["configs"][name]
    smp = sample_pso_oracle(O.load(), c, PSO["gens"], PSO["K"],
                            threads=4 if c["particles"] >= 1024 else 1)
    S.compare(name, c, smp, PSO["gens"])


N4 = S.load_n4()


def n4_start(n):
    return 0.5 + 0.01 * np.arange(n)


@pytest.mark.parametrize("name", ["n2", "n16"])
def test_sann_sync_oracle_vs_reference_distribution(name):
    """SANN: 128 chains of the synchronous oracle (= the device, bit for bit) against
    128 seeded runs of the reference (nlsolver.h:2744-2815): the chain is the same sequential Markov
    chain, only the draws are keyed instead of streamed."""
    c, lib = N4["sann"][name], O.load()
    runs = [O.sann_sync(lib, "rosenbrock", n4_start(c["n"]), S.N4_SEED, k, max_iter=c["max_iter"],
                        temp_iter=c["temperature_iter"], temp_max=c["temperature_max"])[0]
            for k in range(N4["K"])]
    S.compare_n4("sann_" + name, c, [r.f_value for r in runs], [r.iteration for r in runs],
                 [r.function_calls_used for r in runs])


@pytest.mark.parametrize("name", ["n2", "n8"])
def test_nmpso_sync_oracle_vs_reference_distribution(name):
    c, lib = N4["nmpso"][name], O.load()
    runs = [O.nmpso_sync(lib, "rosenbrock", n4_start(c["n"]), S.N4_SEED, k, eps=c["eps"],
                         max_iter=c["max_iter"], no_change=c["no_change"])[0] for k in range(N4["K"])]
    S.compare_n4("nmpso_" + name, c, [r.f_value for r in runs], [r.iteration for r in runs],
                 [r.function_calls_used for r in runs])
