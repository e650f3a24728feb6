"""Throughput of the batched Nelder-Mead engine at small dimensions (run with PYTHONPATH=.)."""
import numpy as np
import nlsolver_amd as m

for n, B in [(2, 1 << 16), (4, 1 << 16), (8, 1 << 15), (16, 1 << 14), (32, 1 << 13), (128, 1 << 12)]:
    rng = np.random.default_rng(n)
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    with m.NMEngine("rosenbrock", B, n, eps=0.0, max_iter=200, no_change_best_tol=10**9) as eng:
        x, st, eps = eng.minimize(x0.copy())
        ms = eng.time_solve(x0, 3) / 3
    iters = sum(s.iteration for s in st)
    print(f"dim {n:4d} starts {B:6d}: {ms:8.3f} ms  {iters / ms * 1e3:.3e} iteration-starts/s", flush=True)
