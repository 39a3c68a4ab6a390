"""ctypes binding of libplinopt_hip.so -- the C-ABI of include/plinopt_hip.h.

This is plumbing only: every compute call goes to the HIP library; if the
library (or a GPU) is missing the calls raise, there is no CPU fallback here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLO_HIP_LIB") or os.environ.get("PLINOPT_HIP_LIB") or os.path.join(_HERE, "libplinopt_hip.so")   # PLO_HIP_LIB: another build of the same library (profiling variants)

PLO_OK = 0
PLO_E_ARG, PLO_E_HIP, PLO_E_CAPACITY, PLO_E_UNSUPPORTED, PLO_E_INTERNAL = -1, -2, -3, -4, -5
COST_SUM_THEN_ADD, COST_ADD_THEN_MUL, COST_SUM, COST_RECSUB = 0, 1, 2, 3
PLAN_HBM = 1

# every symbol include/plinopt_hip.h declares
EXPORTS = [
    "plo_init", "plo_shutdown", "plo_last_error", "plo_device_count",
    "plo_cse_plan_create", "plo_cse_plan_create_ex", "plo_cse_plan_is_hbm", "plo_cse_plan_hbm_counters", "plo_cse_plan_hbm_counters_ex", "plo_cse_plan_destroy",
    "plo_cse_search_plan", "plo_cse_search", "plo_cse_search_multi", "plo_multi_comm_inits", "plo_kernel_search_multi", "plo_tril_search_multi",
    "plo_cse_cost_many_plan", "plo_cse_cost_many",
    "plo_cse_chain_create", "plo_cse_chain_destroy", "plo_cse_chain_search", "plo_cse_chain_cost_many", "plo_cse_chain_batch", "plo_kernel_search",
    "plo_cse_enum_cost_many_plan", "plo_cse_enum_search_plan",
    "plo_cob_search", "plo_cob_search_range", "plo_cob_search_batch",
    "plo_tril_plan_create", "plo_tril_plan_create_x", "plo_tril_plan_create_q", "plo_tril_plan_destroy", "plo_tril_cost_many", "plo_tril_search",
    "plo_pack_cost",
]

u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)


class CSR(ctypes.Structure):
    _fields_ = [("m", ctypes.c_uint32), ("n", ctypes.c_uint32),
                ("rowptr", u32p), ("col", u32p), ("val", u32p)]


class Best(ctypes.Structure):
    _fields_ = [("adds", ctypes.c_uint32), ("muls", ctypes.c_uint32), ("seed", ctypes.c_uint64)]


class Stats(ctypes.Structure):
    _fields_ = [("seconds", ctypes.c_double), ("kernel_ms", ctypes.c_double),
                ("candidates", ctypes.c_uint64), ("launches", ctypes.c_uint32),
                ("lds_bytes", ctypes.c_uint32), ("waves_per_wg", ctypes.c_uint32),
                ("grid", ctypes.c_uint32), ("algo_bytes", ctypes.c_uint64),
                ("reduce", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("reduce_seconds", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CobBest(ctypes.Structure):
    _fields_ = [("zeros_v", ctypes.c_int32), ("zeros_w", ctypes.c_int32), ("index", ctypes.c_uint64), ("found", ctypes.c_uint32)]


class CobProblem(ctypes.Structure):
    """plo_cob_problem_t of include/plinopt_hip.h"""
    _fields_ = [("TM", ctypes.POINTER(ctypes.c_uint32)), ("Cand", ctypes.POINTER(ctypes.c_uint32)), ("coeffs", ctypes.POINTER(ctypes.c_uint32)),
                ("ncoeffs", ctypes.c_uint32), ("p", ctypes.c_uint32), ("w0", ctypes.c_int32), ("w1", ctypes.c_int32)]


class ICSR(ctypes.Structure):
    _fields_ = [("m", ctypes.c_uint32), ("n", ctypes.c_uint32), ("rowptr", u32p), ("col", u32p), ("val", ctypes.POINTER(ctypes.c_int32))]


class QCSR(ctypes.Structure):
    _fields_ = [("m", ctypes.c_uint32), ("n", ctypes.c_uint32), ("rowptr", u32p), ("col", u32p),
                ("num", ctypes.POINTER(ctypes.c_int64)), ("den", ctypes.POINTER(ctypes.c_int64))]


class TrilBest(ctypes.Structure):
    _fields_ = [("add", ctypes.c_uint32), ("sca", ctypes.c_uint32), ("mul", ctypes.c_uint32), ("variant", ctypes.c_uint32), ("seed", ctypes.c_uint64)]


class PloError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libplinopt_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Loads the in-tree HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `make -C plinopt_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so; load it first so
        # that this library binds to the same runtime instead of bringing /opt/rocm's copy beside it
        # (two runtimes in one process do not see each other's device state).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.plo_last_error.restype = ctypes.c_char_p
        L.plo_init.argtypes = [ctypes.c_int]
        L.plo_cse_plan_create.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_cse_plan_create_ex.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.POINTER(ctypes.c_void_p)]
        L.plo_cse_plan_is_hbm.argtypes = [ctypes.c_void_p]
        L.plo_cse_plan_hbm_counters.argtypes = [ctypes.c_void_p, u32p]
        L.plo_cse_plan_hbm_counters_ex.argtypes = [ctypes.c_void_p, u32p, ctypes.c_uint32]
        L.plo_cse_plan_destroy.argtypes = [ctypes.c_void_p]
        L.plo_cse_search_plan.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                          ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_cse_search.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_int, ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_cse_search_multi.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                           ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_cse_cost_many_plan.argtypes = [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p,
                                             ctypes.POINTER(Stats)]
        L.plo_cse_cost_many.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, u64p, ctypes.c_uint64,
                                        ctypes.c_uint64, u32p, u32p]
        L.plo_cse_chain_create.argtypes = [ctypes.POINTER(CSR), ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_cse_chain_destroy.argtypes = [ctypes.c_void_p]
        L.plo_cse_chain_search.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                           ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_cse_chain_cost_many.argtypes = [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p,
                                              ctypes.POINTER(Stats)]
        L.plo_cse_chain_batch.argtypes = [ctypes.c_uint32, ctypes.POINTER(CSR), ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32,
                                          ctypes.c_int, u32p, u32p, ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_kernel_search.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                        u32p, u32p, u32p, ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_cob_search.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.POINTER(CobBest), ctypes.POINTER(Stats)]
        L.plo_cob_search_range.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.POINTER(CobBest), ctypes.POINTER(Stats)]
        L.plo_cob_search_batch.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.POINTER(CobProblem), ctypes.POINTER(CobBest), ctypes.POINTER(Stats)]
        L.plo_cse_enum_cost_many_plan.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p, u64p, ctypes.POINTER(Stats)]
        L.plo_cse_enum_search_plan.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(Best), u64p, ctypes.POINTER(Stats)]
        L.plo_tril_plan_create.argtypes = [ctypes.POINTER(ICSR), ctypes.POINTER(ICSR), ctypes.POINTER(ICSR), ctypes.POINTER(ctypes.c_void_p)]
        L.plo_tril_plan_create_x.argtypes = [ctypes.POINTER(ICSR), ctypes.POINTER(ICSR), ctypes.POINTER(ICSR), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_tril_plan_create_q.argtypes = [ctypes.POINTER(QCSR), ctypes.POINTER(QCSR), ctypes.POINTER(QCSR), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.plo_tril_plan_destroy.argtypes = [ctypes.c_void_p]
        L.plo_tril_plan_destroy.restype = None
        L.plo_tril_cost_many.argtypes = [ctypes.c_void_p, u64p, ctypes.c_uint64, ctypes.c_uint64, u32p, ctypes.POINTER(Stats)]
        L.plo_tril_search.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(TrilBest), ctypes.POINTER(Stats)]
        L.plo_kernel_search_multi.argtypes = [ctypes.POINTER(CSR), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                              ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(Best), ctypes.POINTER(Stats)]
        L.plo_tril_search_multi.argtypes = [ctypes.POINTER(QCSR), ctypes.POINTER(QCSR), ctypes.POINTER(QCSR), ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(TrilBest), ctypes.POINTER(Stats)]
        L.plo_multi_comm_inits.restype = ctypes.c_uint64
        L.plo_pack_cost.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32]
        L.plo_pack_cost.restype = ctypes.c_uint64
        _lib = L
    return _lib


def check(rc):
    if rc != PLO_OK:
        raise PloError(rc, lib().plo_last_error().decode(errors="replace"))


def make_csr(m, n, rowptr, col, val):
    """Returns (CSR struct, keepalive tuple)."""
    rp = (ctypes.c_uint32 * (m + 1))(*rowptr)
    c = (ctypes.c_uint32 * max(len(col), 1))(*col)
    v = (ctypes.c_uint32 * max(len(val), 1))(*val)
    return CSR(m, n, rp, c, v), (rp, c, v)
