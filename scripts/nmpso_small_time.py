"""Throughput of the batched NelderMeadPSO engine at small dimensions (run with PYTHONPATH=.)."""
import numpy as np
import nlsolver_amd as m

for n, B in [(2, 1 << 16), (4, 1 << 16), (8, 1 << 15), (16, 1 << 14), (32, 1 << 12), (64, 1 << 11),
             (128, 1 << 10)]:
    rng = np.random.default_rng(n)
    x0 = 0.5 + (rng.random((B, n)) - 0.5)
    with m.NMPSOEngine("rosenbrock", B, n, eps=0.0, max_iter=100, no_change_best_iter=2**62) as eng:
        x, st = eng.minimize(x0)
        ms = eng.time_solve(x0, 3) / 3
    evals = sum(s.function_calls_used for s in st)
    print(f"dim {n:4d} instances {B:6d}: {ms:8.3f} ms  {evals / ms * 1e3:.3e} evaluations/s  "
          f"{B * 100 / ms * 1e3:.3e} iteration-instances/s", flush=True)
