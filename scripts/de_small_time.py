"""DE turn time at small dimensions (run with PYTHONPATH=.)."""
import time
import numpy as np
import torch
import nlsolver_amd as m

for D in (2, 8, 16, 32, 64, 128):
    pop = 65536
    eng = m.DEEngine("rosenbrock", pop, D, minimize=True, strategy=m.DE_RANDOM, CR=0.9, F=0.8,
                     eps=1e-300, max_iter=10**12, best_val_no_change=10**12, seed=1)
    eng.init(np.full(D, 4.096))
    eng.step(2000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(2000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"D {D:4d} pop {pop}: {dt / 2000 * 1e6:7.2f} us per turn  {pop * 2000 / dt:.3e} candidate-evals/s",
          flush=True)
    eng.close()
