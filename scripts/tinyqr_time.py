"""Kernel time of the batched tinyqr::lm (nlsg_tinyqr_lm) by HIP events, transfers excluded.
usage: python scripts/tinyqr_time.py [batch n p]..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlsolver_amd  # noqa: E402

cases = [(8192, 576, 64), (8192, 64, 64), (8192, 128, 32), (65536, 16, 4)]
if len(sys.argv) > 3:
    a = list(map(int, sys.argv[1:]))
    cases = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
rng = np.random.default_rng(0)
for batch, n, p in cases:
    X = 2 * rng.random((batch, p, n)) - 1
    y = 2 * rng.random((batch, n)) - 1
    nlsolver_amd.tinyqr.lm(X, y)
    ms = min(nlsolver_amd.tinyqr.lm(X, y, return_ms=True)[1] for _ in range(3))
    rot = batch * sum(n - 1 - j for j in range(min(p, n - 1)))
    print(f"batch {batch} n {n} p {p}: {ms:.3f} ms  ({batch / ms * 1e3:.3e} systems/s, "
          f"{rot / ms * 1e3:.3e} rotations/s)", flush=True)
