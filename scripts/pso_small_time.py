"""PSO iteration time at small dimensions (run with PYTHONPATH=.)."""
import time
import numpy as np
import torch
import nlsolver_amd as m

for kind, name in ((m.PSO_ACCELERATED, "accelerated"), (m.PSO_VANILLA, "vanilla")):
    for D in (8, 16, 32, 64, 128):
        n = 131072
        eng = m.PSOEngine("rosenbrock", n, D, type=kind, eps=0.0, max_iter=10**12,
                          best_val_no_change=10**12)
        eng.init(-2.048, 2.048)
        eng.step(300)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(300)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:11s} D {D:4d} n {n}: {dt / 300 * 1e6:7.2f} us per iteration  "
              f"{n * 300 / dt:.3e} particle-evals/s", flush=True)
        eng.close()
