"""Throughput of the packed SANN kernel at small dimensions (run with PYTHONPATH=.)."""
import numpy as np
import nlsolver_amd as m

for n, B in [(2, 1 << 18), (8, 1 << 18), (16, 1 << 17), (32, 1 << 16), (64, 1 << 15), (128, 1 << 14)]:
    rng = np.random.default_rng(n)
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    with m.SANNEngine("rosenbrock", B, n, max_iter=100, temperature_iter=10) as eng:
        eng.minimize(x0)
        ms = eng.time_solve(x0, 3) / 3
    trials = B * 901
    print(f"dim {n:4d} chains {B:7d}: {ms:8.3f} ms  {trials / ms * 1e3:.3e} trial points/s  "
          f"{trials * n / ms * 1e3:.3e} normal draws/s", flush=True)
