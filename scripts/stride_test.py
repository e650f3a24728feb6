import sys, time
import numpy as np
import nlsolver_amd
rng = np.random.default_rng(1)
for n in (1024, 1008, 992):
    d = np.array([1.0 + 9.0 * i / (n - 1) for i in range(n)])
    b = np.sin(0.1 * np.arange(n))
    xq = 1.0 + 0.5 * (rng.random((512, n)) - 0.5)
    for ref in (False, True):
        with nlsolver_amd.BFGSEngine(nlsolver_amd.QuadDiagRank1(d, b, 0.01), 512, max_iter=20, grad_eps=0.0, reference_order=ref) as eng:
            eng.minimize(xq.copy())
            t0 = time.perf_counter()
            for _ in range(3):
                eng.minimize(xq.copy())
            print(n, ref, (time.perf_counter() - t0) / 3 * 1e3, "ms", flush=True)
