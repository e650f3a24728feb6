"""nlsolver_amd — MI355X-native iteration engine behind the nlsolver API surface.

Layout:
    csrc/                 HIP kernels for gfx950 + the extern "C" boundary
    libnlsolver_hip.so    built in-tree by `make -C nlsolver_amd/csrc`
    _capi.py              ctypes binding of include/nlsg_c_api.h
    de.py                 DE / DESolver: mirror of nlsolver::DE (nlsolver.h:2379-2477)
    pso.py                PSO / PSOSolver: mirror of nlsolver::PSO (nlsolver.h:2498-2742)
    bfgs.py               BFGS (batched starts): mirror of nlsolver::BFGS (nlsolver.h:3169-3286)
    lm.py                 LevenbergMarquardt (batched NLLS): mirror of nlsolver.h:3428-3545
    nm.py                 NelderMead (batched starts): mirror of nlsolver.h:2099-2300
    sann.py               SANN (batched chains): mirror of nlsolver.h:2744-2815
    nmpso.py              NelderMeadPSO (batched instances): mirror of nlsolver.h:3546-3920
    tinyqr.py             tinyqr.lm on batches of n x p systems: mirror of tinyqr.h:461-470
    dist.py               population sharding across ranks (torch.distributed / RCCL)
"""
from ._capi import DE_BEST, DE_RANDOM, PSO_ACCELERATED, PSO_VANILLA, NlsgError, pinned_empty  # noqa: F401
from .de import DE, CustomObjective, DEEngine, DESolver  # noqa: F401
from .pso import PSO, PSOEngine, PSOSolver  # noqa: F401
from .bfgs import BFGS, BFGSEngine, QuadDiagRank1  # noqa: F401
from .lm import LevenbergMarquardt, LMEngine, TanhRegression  # noqa: F401
from .nm import NelderMead, NMEngine  # noqa: F401
from .sann import SANN, SANNEngine  # noqa: F401
from .nmpso import NelderMeadPSO, NMPSOEngine  # noqa: F401
from . import tinyqr  # noqa: F401
