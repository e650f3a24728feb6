"""Per-iteration time of the LM engine past 128 parameters, blocked matrix-core step against the
column-by-column step (NLSG_LM_WIDE_CHOL=0). Run with PYTHONPATH=.; HIP events, data resident."""
import os
import sys

import numpy as np

import nlsolver_amd as m

rng = np.random.default_rng(1)
cases = [(1024, 256, 1024, 6), (512, 192, 1024, 6), (1024, 512, 128, 4), (256, 1024, 32, 2)]
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for (mm, n, B, iters) in cases:
    A = (2 * rng.random((B, mm, n)) - 1) / np.sqrt(n)
    star = 2 * rng.random((B, n)) - 1
    y = np.tanh(np.einsum("bmn,bn->bm", A, star))
    th0 = 0.5 * star + 0.05 * (2 * rng.random((B, n)) - 1)
    res = {}
    for sw in ("1", "0"):
        os.environ["NLSG_LM_WIDE_CHOL"] = sw
        with m.LMEngine(m.TanhRegression(A, y), lam=10.0, max_iter=iters, f_delta=0.0) as eng:
            th, st, lam = eng.minimize(th0.copy())
            eng.time_solve(th0, 1)
            dt = eng.time_solve(th0, 2) / 2  # ms, the whole solve: iters + 1 evaluations, iters steps
            ev = eng.time_eval_kernel(th0, 5) / 5
        res[sw] = (dt / iters, ev, th, (dt - (iters + 1) * ev) / iters)
    same = np.array_equal(res["1"][2], res["0"][2], equal_nan=True)
    print(f"LM tanh m={mm} n={n} batch={B}: evaluation {res['1'][1]:.3f} ms; solve / iterations "
          f"{res['1'][0]:.3f} ms blocked vs {res['0'][0]:.3f} ms column-wise "
          f"(step ~ {res['1'][3]:.3f} vs {res['0'][3]:.3f}); same bits: {same}",
          flush=True)
