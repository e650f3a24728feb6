"""ctypes binding of libnlsolver_hip.so (include/nlsg_c_api.h).

The library is the product's only compute path. There is no CPU fallback: if
the shared object is missing or no gfx950 device is visible, calls raise.
"""
import ctypes as C
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# $NLSG_LIBRARY names another build of the library (as for the C++ header): A/B runs of variants
LIB_PATH = os.environ.get("NLSG_LIBRARY") or os.path.join(HERE, "libnlsolver_hip.so")

u64, i32, f64 = C.c_uint64, C.c_int32, C.c_double
pd = C.POINTER(C.c_double)
pu = C.POINTER(C.c_uint64)

OBJECTIVES = {"rosenbrock": 0, "sphere": 1, "styblinski_tang": 2, "rastrigin": 3}
DE_BEST, DE_RANDOM = 0, 1  # enum RecombinationStrategy { best, random }, nlsolver.h:2377


class NlsgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nlsg error {code}: {msg}")
        self.code = code


class Status(C.Structure):
    """nlsg_status == solver_status<T> (nlsolver.h:2054-2097) + engine fields."""
    _fields_ = [("f_value", f64), ("iteration", u64), ("function_calls_used", u64),
                ("gradient_evals_used", u64), ("hessian_evals_used", u64),
                ("best_index", u64), ("val_no_change", u64), ("std_err", f64),
                ("done", i32), ("reserved", i32)]

    def get_summary(self):
        # tuple order of solver_status::get_summary, nlsolver.h:2079-2083
        return (self.function_calls_used, self.iteration, self.f_value,
                self.gradient_evals_used, self.hessian_evals_used)


class DEConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("minimize", i32), ("strategy", i32), ("trace", i32),
                ("pop", u64), ("dim", u64), ("shard_lo", u64), ("shard_n", u64),
                ("CR", f64), ("F", f64), ("eps", f64),
                ("max_iter", u64), ("best_val_no_change", u64), ("seed", u64)]


OBJ_CUSTOM = 64


class CustomObjectiveC(C.Structure):  # nlsg_custom_objective
    _fields_ = [("term_body", C.c_char_p), ("finish_body", C.c_char_p), ("chain", i32),
                ("reserved", i32)]


PSO_VANILLA, PSO_ACCELERATED = 0, 1  # enum PSOType { Vanilla, Accelerated }, nlsolver.h:2496


class PSOConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("minimize", i32), ("type", i32), ("bounded", i32),
                ("n_particles", u64), ("dim", u64), ("shard_lo", u64), ("shard_n", u64),
                ("inertia", f64), ("cognitive", f64), ("social", f64), ("eps", f64),
                ("max_iter", u64), ("best_val_no_change", u64), ("seed", u64)]


OBJ_QUAD_DIAG_RANK1 = 16


class BFGSConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("flags", i32), ("batch", u64), ("dim", u64),
                ("max_iter", u64), ("grad_eps", f64), ("alpha", f64), ("quad_c", f64)]


BFGS_SYMMETRIC = 1  # NLSG_BFGS_SYMMETRIC
BFGS_REFERENCE_ORDER = 2  # NLSG_BFGS_REFERENCE_ORDER


OBJ_TANH_REGRESSION = 32
LM_CHOLESKY, LM_QR, LM_CHOLESKY_REFERENCE_ORDER = 0, 1, 2


class LMConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("solver", i32), ("batch", u64), ("m", u64), ("n", u64),
                ("lambda_", f64), ("up", f64), ("down", f64), ("max_iter", u64), ("f_delta", f64)]


NM_REFERENCE_ORDER = 1  # NLSG_NM_REFERENCE_ORDER


class NMConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("minimize", i32), ("bounded", i32), ("flags", i32),
                ("batch", u64), ("dim", u64),
                ("step", f64), ("alpha", f64), ("gamma", f64), ("rho", f64), ("sigma", f64),
                ("eps", f64), ("max_iter", u64), ("no_change_best_tol", u64), ("restarts", u64)]


class SANNConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("minimize", i32), ("batch", u64), ("dim", u64),
                ("chain_lo", u64), ("max_iter", u64), ("temperature_iter", u64),
                ("temperature_max", f64), ("seed", u64)]


class NMPSOConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", i32), ("stream", C.c_void_p),
                ("objective", i32), ("minimize", i32), ("bounded", i32), ("reserved", i32),
                ("batch", u64), ("dim", u64), ("inst_lo", u64),
                ("alpha", f64), ("gamma", f64), ("rho", f64), ("sigma", f64),
                ("inertia", f64), ("cognitive", f64), ("social", f64), ("eps", f64),
                ("max_iter", u64), ("no_change_best_iter", u64), ("seed", u64)]


# every symbol include/nlsg_c_api.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "nlsg_last_error": (C.c_char_p, []),
    "nlsg_abi_version": (C.c_int, []),
    "nlsg_device_count": (C.c_int, []),
    "nlsg_call_timing": (C.c_int, [pd]),
    "nlsg_release_cached": (C.c_int, []),
    "nlsg_cached_bytes": (u64, []),
    "nlsg_probe_math": (C.c_int, [C.c_int32, pu, pu, u64, C.c_int32]),
    "nlsg_de_create": (C.c_int, [C.POINTER(DEConfig), C.POINTER(_H)]),
    "nlsg_de_destroy": (C.c_int, [_H]),
    "nlsg_de_init": (C.c_int, [_H, pd]),
    "nlsg_de_step": (C.c_int, [_H, u64]),
    "nlsg_de_minimize": (C.c_int, [_H, pd, u64, C.POINTER(Status)]),
    "nlsg_de_status": (C.c_int, [_H, C.POINTER(Status)]),
    "nlsg_de_best": (C.c_int, [_H, pd, pd, pu]),
    "nlsg_de_download": (C.c_int, [_H, pd, pd, pu]),
    "nlsg_de_upload": (C.c_int, [_H, pd, pd]),
    "nlsg_de_time_generation_kernel": (C.c_int, [_H, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_de_time_turns": (C.c_int, [_H, u64, C.POINTER(C.c_float)]),
    "nlsg_de_record_doubles": (u64, [_H]),
    "nlsg_de_turn_begin": (C.c_int, [_H, C.c_void_p]),
    "nlsg_de_turn_end": (C.c_int, [_H, C.c_void_p, i32]),
    "nlsg_de_turn_finalize": (C.c_int, [_H, C.c_void_p, i32]),
    "nlsg_de_turn_generation": (C.c_int, [_H]),
    "nlsg_de_can_speculate": (C.c_int, [_H]),
    "nlsg_rtc_load": (C.c_int, [C.c_char_p]),
    "nlsg_de_create_custom": (C.c_int, [C.POINTER(DEConfig), C.POINTER(CustomObjectiveC),
                                        C.POINTER(C.c_void_p)]),
    "nlsg_pso_create_custom": (C.c_int, [C.POINTER(PSOConfig), C.POINTER(CustomObjectiveC),
                                         C.POINTER(C.c_void_p)]),
    "nlsg_nm_create_custom": (C.c_int, [C.POINTER(NMConfig), C.POINTER(CustomObjectiveC),
                                        C.POINTER(C.c_void_p)]),
    "nlsg_bfgs_create_custom": (C.c_int, [C.POINTER(BFGSConfig), C.POINTER(CustomObjectiveC),
                                          C.POINTER(C.c_void_p)]),
    "nlsg_lm_create_custom": (C.c_int, [C.POINTER(LMConfig), C.POINTER(CustomObjectiveC),
                                        C.POINTER(C.c_void_p)]),
    "nlsg_comm_load": (C.c_int, [C.c_char_p]),
    "nlsg_comm_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "nlsg_de_comm_attach": (C.c_int, [_H, C.POINTER(C.c_ubyte), C.c_int32, C.c_int32]),
    "nlsg_de_step_sharded": (C.c_int, [_H, C.c_uint64]),
    "nlsg_de_comm_ranks": (C.c_int, [_H, C.POINTER(i32), C.POINTER(i32)]),
    "nlsg_pso_comm_ranks": (C.c_int, [_H, C.POINTER(i32), C.POINTER(i32)]),
    "nlsg_pso_comm_attach": (C.c_int, [_H, C.POINTER(C.c_ubyte), C.c_int32, C.c_int32]),
    "nlsg_pso_step_sharded": (C.c_int, [_H, C.c_uint64]),
    "nlsg_pso_create": (C.c_int, [C.POINTER(PSOConfig), C.POINTER(_H)]),
    "nlsg_pso_destroy": (C.c_int, [_H]),
    "nlsg_pso_init": (C.c_int, [_H, pd, pd]),
    "nlsg_pso_step": (C.c_int, [_H, u64]),
    "nlsg_pso_minimize": (C.c_int, [_H, pd, pd, pd, u64, C.POINTER(Status)]),
    "nlsg_pso_status": (C.c_int, [_H, C.POINTER(Status)]),
    "nlsg_pso_best": (C.c_int, [_H, pd, pd, pu]),
    "nlsg_pso_download": (C.c_int, [_H, pd, pd, pd, pd]),
    "nlsg_pso_time_move_kernel": (C.c_int, [_H, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_pso_record_doubles": (u64, [_H]),
    "nlsg_pso_turn_begin": (C.c_int, [_H, C.c_void_p]),
    "nlsg_pso_turn_end": (C.c_int, [_H, C.c_void_p, i32]),
    "nlsg_bfgs_create": (C.c_int, [C.POINTER(BFGSConfig), pd, pd, C.POINTER(_H)]),
    "nlsg_bfgs_destroy": (C.c_int, [_H]),
    "nlsg_bfgs_init": (C.c_int, [_H, pd]),
    "nlsg_bfgs_step": (C.c_int, [_H, u64]),
    "nlsg_bfgs_unfinished": (C.c_int, [_H, pu]),
    "nlsg_bfgs_identity_count": (C.c_int, [_H, pu]),
    "nlsg_bfgs_download": (C.c_int, [_H, pd, C.POINTER(Status)]),
    "nlsg_bfgs_download_state": (C.c_int, [_H, pd, pd]),
    "nlsg_bfgs_minimize": (C.c_int, [_H, pd, C.POINTER(Status)]),
    "nlsg_bfgs_time_steps": (C.c_int, [_H, u64, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "nlsg_tinyqr_lm": (C.c_int, [pd, pd, u64, u64, u64, f64, i32, pd, C.POINTER(C.c_float)]),
    "nlsg_tinyqr_lm_device": (C.c_int, [C.c_void_p, C.c_void_p, u64, u64, u64, f64, i32, C.c_void_p,
                                        C.c_void_p]),
    "nlsg_tinyqr_qr": (C.c_int, [pd, pd, u64, u64, u64, f64, i32, pd, pd, pd]),
    "nlsg_lm_create": (C.c_int, [C.POINTER(LMConfig), C.POINTER(_H)]),
    "nlsg_lm_destroy": (C.c_int, [_H]),
    "nlsg_lm_set_data": (C.c_int, [_H, pd, pd]),
    "nlsg_lm_set_solver": (C.c_int, [_H, i32]),
    "nlsg_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), u64]),
    "nlsg_host_free": (C.c_int, [C.c_void_p]),
    "nlsg_lm_minimize": (C.c_int, [_H, pd, C.POINTER(Status), pd]),
    "nlsg_lm_time_solve": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_lm_time_eval_kernel": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_lm_time_qr_kernel": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_nm_create": (C.c_int, [C.POINTER(NMConfig), C.POINTER(_H)]),
    "nlsg_nm_destroy": (C.c_int, [_H]),
    "nlsg_nm_minimize": (C.c_int, [_H, pd, pd, pd, C.POINTER(Status), pd]),
    "nlsg_nm_phase_cycles": (C.c_int, [_H, pd, pu]),
    "nlsg_nm_time_solve": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_nmpso_create": (C.c_int, [C.POINTER(NMPSOConfig), C.POINTER(_H)]),
    "nlsg_nmpso_create_custom": (C.c_int, [C.POINTER(NMPSOConfig), C.POINTER(CustomObjectiveC),
                                           C.POINTER(C.c_void_p)]),
    "nlsg_nmpso_destroy": (C.c_int, [_H]),
    "nlsg_nmpso_minimize": (C.c_int, [_H, pd, pd, pd, C.POINTER(Status)]),
    "nlsg_nmpso_time_solve": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
    "nlsg_sann_create": (C.c_int, [C.POINTER(SANNConfig), C.POINTER(_H)]),
    "nlsg_sann_create_custom": (C.c_int, [C.POINTER(SANNConfig), C.POINTER(CustomObjectiveC),
                                          C.POINTER(C.c_void_p)]),
    "nlsg_sann_destroy": (C.c_int, [_H]),
    "nlsg_sann_minimize": (C.c_int, [_H, pd, C.POINTER(Status)]),
    "nlsg_sann_time_solve": (C.c_int, [_H, pd, C.c_uint32, C.POINTER(C.c_float)]),
}

_lib = None


def lib():
    """Load libnlsolver_hip.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NlsgError(-1, f"{LIB_PATH} is missing: build it with "
                                "`make -C nlsolver_amd/csrc` (no CPU fallback exists)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise NlsgError(rc, lib().nlsg_last_error().decode(errors="replace"))


class _PinnedBlock:
    """Owner of one nlsg_host_alloc block; numpy arrays made by pinned_empty keep it alive."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        check(lib().nlsg_host_alloc(C.byref(self.ptr), nbytes))

    def __del__(self):
        # at interpreter shutdown the module globals, the ctypes handle or the HIP runtime may be
        # gone already: the process is about to release the memory anyway
        try:
            if sys.is_finalizing():
                return
            if getattr(self, "ptr", None) and self.ptr.value:
                lib().nlsg_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype="float64"):
    """numpy array in page-locked host memory (nlsg_host_alloc): what is handed to the engines from
    there crosses PCIe by DMA at the link's rate (e.g. the design matrices of an LM model). The
    memory is released when the array (and every view of it) is gone."""
    import numpy as np
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dt.itemsize, 8)
    block = _PinnedBlock(nbytes)
    buf = (C.c_char * nbytes).from_address(block.ptr.value)
    buf._nlsg_owner = block  # the array's base is `buf`; `buf` keeps the block alive
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


PROBE_FUNCTIONS = {"log": 0, "cos": 1, "exp": 2, "tanh": 3, "cos_2pi": 4, "u01": 5, "rnorm": 6}


def probe_math(fn, bits, device=0):
    """nlsg_probe_math: the device's deterministic primitive `fn` ("log", "cos", "exp", "tanh",
    "cos_2pi": argument = a double's bit pattern; "u01", "rnorm": argument = a 64-bit draw) on
    a uint64 array; returns the results' bit patterns."""
    import numpy as np
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    out = np.empty_like(bits)
    check(lib().nlsg_probe_math(PROBE_FUNCTIONS[fn], bits.ctypes.data_as(pu), out.ctypes.data_as(pu),
                                bits.size, device))
    return out
