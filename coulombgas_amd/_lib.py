"""ctypes binding of include/coulombgas.h (libcoulombgas_hip.so).

There is deliberately NO fallback: if the HIP library is missing or cannot be loaded the
import of any compute entry point raises.  (oracle/ and tests/host_emul are test
infrastructure and are never loaded from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COULOMBGAS_HIP_LIB", os.path.join(_HERE, "lib", "libcoulombgas_hip.so"))   # override: A/B builds of the same ABI

CG_OK, CG_ERR_ARG, CG_ERR_HIP, CG_ERR_UNSUPPORTED, CG_ERR_STATE, CG_ERR_RCCL = 0, -1, -2, -3, -4, -5
CG_PTR_HOST, CG_PTR_DEVICE = 0, 1
CG_LAP_EXACT, CG_LAP_HUTCHINSON, CG_LAP_HUTCHINSON_SPLIT = 0, 1, 2


class CoulombGasError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("coulombgas_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)

# name -> (restype, argtypes); every symbol include/coulombgas.h declares
PROTOTYPES = {
    "cg_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int]),
    "cg_destroy": (None, [C.c_void_p]),
    "cg_last_error": (C.c_char_p, [C.c_void_p]),
    "cg_set_pointer_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "cg_sync": (C.c_int, [C.c_void_p]),
    "cg_num_params": (C.c_int, [C.c_void_p]),
    "cg_set_flow_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cg_set_ewald": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_double]),
    "cg_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cg_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cg_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cg_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cg_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "cg_timer_start": (C.c_int, [C.c_void_p]),
    "cg_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "cg_set_block_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "cg_get_launch_info": (C.c_int, [C.c_void_p, c_lp]),
    "cg_microbench_fp64": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "cg_flow_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_flow_jacobian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_logpsi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_logphi_logjacdet": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cg_logp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_mcmc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64,
                          C.c_void_p, C.c_void_p, C.c_void_p, c_lp]),
    "cg_mcmc_accepts": (C.c_int, [C.c_void_p, c_lp]),
    "cg_mcmc_accept_rate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_double)]),
    "cg_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "cg_ewald": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_grad_laplacian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_param_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_quantum_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_quantum_fisher": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cg_scores_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "cg_grad_laplacian_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_scores_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_scores_fisher": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_scores_mean": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cg_van_num_params": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cg_van_set_params": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cg_van_log_prob": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cg_van_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_van_scores_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "cg_van_scores_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_van_scores_fisher": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_van_scores_get": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cg_local_energy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_abs_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cg_clip_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "cg_randn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64]),
    "cg_axpby": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_size_t]),
    "cg_scale_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double]),
    "cg_fisher_real": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cg_cholesky": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "cg_spd_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cg_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cg_comm_destroy": (None, [C.c_void_p]),
    "cg_allreduce_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cg_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
}


def lib():
    """Loads libcoulombgas_hip.so (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -m coulombgas_amd.build` "
                              "(hipcc --offload-arch=gfx950); coulombgas_amd has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().cg_last_error(ctx)
        raise CoulombGasError(rc, msg.decode() if msg else "?")
