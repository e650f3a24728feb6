"""Where a Nelder-Mead iteration's time goes: the phase counters of both
kernels (nlsg_nm_phase_cycles) on the bench workload — Rosenbrock-128D, 2000 iterations — for a
batch that fills the chip once (one workgroup per CU) and the whole-solve times of both kernels.
usage: python scripts/nm_phases.py [out.json]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlsolver_amd  # noqa: E402

NAMES = ["scan (std_err, best/worst/second, stop tests)", "centroid", "reflection (transform, evaluation, decision)",
         "expansion / contraction", "shrink + rescoring"]
n, iters = 128, 2000
out = {"workload": f"Nelder-Mead Rosenbrock-{n}D, {iters} iterations per start, eps = 0"}
for sw, name in (("0", "nm_solve_kernel (a workgroup phase per link of the decision chain)"),
                 ("1", "nm_solve_driver_kernel (one wave drives, the others shrink)")):
    os.environ["NLSG_NM_DRIVER"] = sw
    entry = {}
    for batch in (256, 4096):
        rng = np.random.default_rng(7)
        x0 = 0.5 + 0.2 * (rng.random((batch, n)) - 0.5)
        with nlsolver_amd.NMEngine("rosenbrock", batch, n, eps=0.0, max_iter=iters,
                                   no_change_best_tol=10**9) as eng:
            eng.time_solve(x0, 1)
            ms = min(eng.time_solve(x0, 1) for _ in range(2))
            x, st, _ = eng.minimize(x0.copy())
            cyc = eng.phase_cycles(x0).astype(np.float64)
        tot = cyc[:, :5].sum(axis=1)
        entry[f"batch_{batch}"] = {
            "solve_ms": ms, "us_per_iteration_per_round": ms * 1e3 / iters / max(1, batch // 256),
            "iteration_starts_per_s": batch * iters / (ms * 1e-3),
            "shrinks_per_iteration": float((cyc[:, 7] / cyc[:, 6]).mean()),
            "objective_calls_per_iteration": float(np.mean([s.function_calls_used for s in st]) / iters),
            "phase_share": {NAMES[k]: float((cyc[:, k] / tot).mean()) for k in range(5)},
            "phase_cycles_per_iteration": {NAMES[k]: float((cyc[:, k] / cyc[:, 6]).mean()) for k in range(5)},
            "checksum_f": float(sum(s.f_value for s in st))}
    out[name] = entry
    print(name, json.dumps(entry, indent=1), flush=True)
a, b = [out[k]["batch_4096"]["checksum_f"] for k in out if k != "workload"]
out["same_results"] = a == b
print("same results:", a == b)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
