"""Where a Nelder-Mead iteration's cycles go, tree order against reference order (Rosenbrock-128D, 256
starts = one workgroup per CU, 2000 iterations): the kernel's phase counters, mean cycles per iteration.
PYTHONPATH=. python scripts/nm_phases.py"""
import numpy as np

import nlsolver_amd

n, iters, batch = 128, 2000, 256
rng = np.random.default_rng(7)
x0 = 0.5 + 0.2 * (rng.random((batch, n)) - 0.5)
names = ["scan", "centroid", "reflection", "expand/contract", "shrink", "-", "iterations", "shrinks"]
for ref in (False, True):
    with nlsolver_amd.NMEngine("rosenbrock", batch, n, eps=0.0, max_iter=iters, no_change_best_tol=10**9,
                               reference_order=ref) as eng:
        ph = eng.phase_cycles(x0).astype(np.float64)
    it = ph[:, 6].mean()
    print(f"reference_order={ref}: iterations {it:.0f}, shrinks {ph[:, 7].mean():.0f}; cycles per iteration: " +
          ", ".join(f"{names[k]} {ph[:, k].mean() / it:.0f}" for k in range(5)))
