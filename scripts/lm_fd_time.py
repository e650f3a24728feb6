import time, numpy as np, nlsolver_amd as m
for obj, n, B, it in [("rosenbrock", 16, 8192, 5), ("rosenbrock", 4, 65536, 5), ("styblinski_tang", 64, 2048, 2), ("rosenbrock", 32, 4096, 3)]:
    rng = np.random.default_rng(1)
    x0 = 0.8 + 0.4 * (rng.random((B, n)) - 0.5)
    with m.lm.LMEngine(obj, batch=B, n=n, lam=10.0, max_iter=it, f_delta=0.0) as eng:
        eng.minimize(x0.copy())
        ms = eng.time_solve(x0, 3) / 3
    evals = B * (it + 1) * (1 + 4 * n + 16 * n * n)
    print(obj, n, B, it, "ms/solve %.3f" % ms, "probe evals/s %.3e" % (evals / ms * 1e3), "iter-problems/s %.3e" % (B * it / ms * 1e3), flush=True)
