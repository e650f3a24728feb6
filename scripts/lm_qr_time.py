"""QR step kernel alone at several batch sizes (n = 64): one round of workgroups per CU vs many."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nlsolver_amd
from nlsolver_amd import _capi

m, n = 64, 64
for batch in (256, 512, 1024, 2048, 8192):
    rng = np.random.default_rng(1)
    A = (2 * rng.random((batch, m, n)) - 1) / np.sqrt(n)
    star = 2 * rng.random((batch, n)) - 1
    y = np.tanh(np.einsum("bmn,bn->bm", A, star))
    t0 = 0.5 * star + 0.1 * (2 * rng.random((batch, n)) - 1)
    with nlsolver_amd.LMEngine(nlsolver_amd.TanhRegression(A, y), lam=10.0, max_iter=20, f_delta=0.0,
                               solver=_capi.LM_QR) as eng:
        eng.time_qr_kernel(t0, 3)
        ms = eng.time_qr_kernel(t0, 10) / 10
    print(f"batch {batch:5d}: {ms * 1e3:8.1f} us per QR step launch, {ms * 1e3 / max(1, batch / 1024):8.1f} us per 1024 problems")
