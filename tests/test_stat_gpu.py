"""Distribution-level parity of the DEVICE engines against the reference itself (tests/_stat.py):
K = 128 device runs per configuration (distinct counter seeds, through the C-ABI) against the
K = 128 seeded runs of the unmodified reference in tests/golden/{de,pso}_stat.json.

Replaces the former single-sample checks ("within 0.3x-3x of the reference's 45 iterations",
"f < 0.5"). What is asserted per statistic — a two-sample KS test, or a band on the median ratio
where the synchronous generation is measurably slower than the reference's in-place one — is
defined once, in tests/_stat.py, and applied to the CPU oracle too (tests/test_stat_oracle.py).
The per-statistic p-values / ratios of the run are written to gpurun_out/stat_report.json.
"""
import json
import os

import numpy as np
import pytest

from tests import _stat as S

pytestmark = pytest.mark.gpu

DE = S.load("de_stat.json")
PSO = S.load("pso_stat.json")
REPORT = {}


@pytest.fixture(scope="module")
def mod():
    import nlsolver_amd
    return nlsolver_amd


@pytest.fixture(scope="module", autouse=True)
def write_report():
    yield
    out = os.path.join(S.HERE, "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "stat_report.json"), "w") as fh:
        json.dump(REPORT, fh, indent=1)


@pytest.mark.parametrize("name", list(DE["configs"]))
def test_de_device_vs_reference_distribution(mod, name):
    c = DE["configs"][name]
    gens, smp = DE["gens"], S.Sample(DE["gens"])
    for k in range(DE["K"]):
        with mod.DEEngine("rosenbrock", c["pop"], c["D"],
                          strategy=mod.DE_RANDOM if c["strategy"] == "random" else mod.DE_BEST,
                          eps=c["eps"], CR=c["CR"], F=c["F"], max_iter=c["max_iter"],
                          best_val_no_change=c["no_change"], seed=S.seed_of(k)) as eng:
            eng.init(S.x0_of(c))

            def snapshot():
                scores = eng.download()[1]
                return scores.min(), scores.mean()

            def finish():
                st = eng.status()
                assert st.done == 1
                return int(st.iteration), float(st.f_value)

            S.run_marks(gens, c["max_iter"], eng.step, snapshot, finish, smp)
    S.compare(name, c, smp.arrays(), gens, REPORT)


@pytest.mark.parametrize("name", list(PSO["configs"]))
def test_pso_device_vs_reference_distribution(mod, name):
    c = PSO["configs"][name]
    gens, smp = PSO["gens"], S.Sample(PSO["gens"])
    x0 = S.x0_of(c)
    for k in range(PSO["K"]):
        with mod.PSOEngine("rosenbrock", c["particles"], c["D"], type=mod.PSO_ACCELERATED,
                           bounded=False, eps=c["eps"], max_iter=c["max_iter"],
                           best_val_no_change=c["no_change"], seed=S.seed_of(k)) as eng:
            eng.init(-np.abs(x0), np.abs(x0))  # nlsolver.h:2553-2560

            def snapshot():
                pbest = eng.download()[2]
                return pbest.min(), pbest.mean()

            def finish():
                st = eng.status()
                assert st.done == 1
                return int(st.iteration), float(st.f_value)

            S.run_marks(gens, c["max_iter"], eng.step, snapshot, finish, smp)
    S.compare(name, c, smp.arrays(), gens, REPORT)


def test_device_sample_is_the_oracle_sample(mod):
    """The device runs above ARE the synchronous oracle's runs (bit for bit): one seed of one
    configuration re-checked here, so the CPU leg's verdicts carry over to the device."""
    from tests import _oracle as O
    c = DE["configs"]["random_pop256_D16"]
    k = 17
    ref = O.DESyncRun(O.load(), "rosenbrock", c["pop"], c["D"], S.x0_of(c), eps=c["eps"],
                      max_iter=c["max_iter"], best_val_no_change=c["no_change"], seed=S.seed_of(k))
    while not ref.s.done:
        ref.step()
    with mod.DEEngine("rosenbrock", c["pop"], c["D"], eps=c["eps"], max_iter=c["max_iter"],
                      best_val_no_change=c["no_change"], seed=S.seed_of(k)) as eng:
        eng.init(S.x0_of(c))
        eng.step(c["max_iter"] + 1)
        st = eng.status()
        scores = eng.download()[1]
    assert int(st.iteration) == int(ref.s.iter) and np.array_equal(scores, ref.scores)


N4 = S.load_n4()


@pytest.mark.parametrize("name", list(N4["sann"]))
def test_sann_device_vs_reference_distribution(mod, name):
    """128 device chains (one engine call, chain k keyed by (seed, k)) against 128
    seeded runs of the reference's SANN (tests/golden/n4_stat.json)."""
    c, K = N4["sann"][name], N4["K"]
    x0 = np.tile(0.5 + 0.01 * np.arange(c["n"]), (K, 1))
    with mod.SANNEngine("rosenbrock", K, c["n"], max_iter=c["max_iter"],
                        temperature_iter=c["temperature_iter"], temperature_max=c["temperature_max"],
                        seed=S.N4_SEED) as eng:
        _, st = eng.minimize(x0)
    S.compare_n4("sann_" + name, c, [s.f_value for s in st], [s.iteration for s in st],
                 [s.function_calls_used for s in st], REPORT)


@pytest.mark.parametrize("name", list(N4["nmpso"]))
def test_nmpso_device_vs_reference_distribution(mod, name):
    c, K = N4["nmpso"][name], N4["K"]
    x0 = np.tile(0.5 + 0.01 * np.arange(c["n"]), (K, 1))
    with mod.NMPSOEngine("rosenbrock", K, c["n"], eps=c["eps"], max_iter=c["max_iter"],
                         no_change_best_iter=c["no_change"], seed=S.N4_SEED) as eng:
        _, st = eng.minimize(x0)
    S.compare_n4("nmpso_" + name, c, [s.f_value for s in st], [s.iteration for s in st],
                 [s.function_calls_used for s in st], REPORT)
