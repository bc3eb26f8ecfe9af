"""ctypes binding of libcapital_hip.so (include/capital_hip.h) -- the product's C-ABI.

There is NO CPU fallback: importing is cheap, but creating a Handle without the built
library or without a GPU raises.  torch is imported first so that this process maps
one HIP runtime (torch's libamdhip64.so.7 satisfies the library's dependency).
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (must precede loading the HIP library)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPITAL_HIP_LIB", os.path.join(_HERE, "libcapital_hip.so"))   # override: A/B builds of the same ABI

NOTRANS, TRANS = 0, 1
LEFT, RIGHT = 0, 1
LOWER, UPPER = 0, 1
NONUNIT, UNIT = 0, 1
RECT, UPPERTRI, LOWERTRI = 0, 1, 2

_i64, _dbl, _int, _vp = C.c_int64, C.c_double, C.c_int, C.c_void_p

# name -> argtypes after the handle (None = no handle argument)
_SIGS = {
    "capi_dgemm": [_int, _int, _i64, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64],
    "capi_dsyrk": [_int, _int, _i64, _i64, _dbl, _vp, _i64, _dbl, _vp, _i64],
    "capi_dgemmt": [_int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64],
    "capi_dsyrk_panel32": [_i64, _i64, _dbl, _vp, _dbl, _vp, _i64],
    "capi_dtrmm_right_panel32": [_i64, _i64, _dbl, _vp, _i64, _vp, _i64, _vp, _i64],
    "capi_dtrmm": [_int, _int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64],
    "capi_dtrmm_oop": [_int, _int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _vp, _i64],
    "capi_dtrmm_acc": [_int, _int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64],
    "capi_dtrsm": [_int, _int, _int, _int, _i64, _i64, _dbl, _vp, _i64, _vp, _i64],
    "capi_dpotrf": [_int, _i64, _vp, _i64],
    "capi_dtrtri": [_int, _int, _i64, _vp, _i64],
    "capi_dpotrf_trtri": [_i64, _vp, _i64, _vp, _i64],
    "capi_get_info": [C.POINTER(_int)],
    "capi_reset_info": [],
    "capi_serialize": [_int, _int, _vp, _i64, _i64, _vp, _i64, _i64] + [_i64] * 8,
    "capi_serialize_shape": [_int, _int, _int, _vp, _i64, _i64, _vp, _i64, _i64] + [_i64] * 8,
    "capi_dlacpy": [_int, _i64, _i64, _vp, _i64, _vp, _i64],
    "capi_dgeadd": [_int, _i64, _i64, _dbl, _vp, _i64, _dbl, _vp, _i64],
    "capi_dtrizero": [_int, _i64, _vp, _i64],
    "capi_daxpby": [_i64, _dbl, _vp, _vp],
    "capi_remove_triangle": [C.c_char, _vp, _i64, _i64, _i64, _i64, _i64],
    "capi_block_to_cyclic": [_vp, _vp, _i64, _i64, _i64],
    "capi_block_to_cyclic_full": [_vp, _vp, _i64, _i64, _i64],
    "capi_cyclic_to_block": [_vp, _vp, _i64, _i64, _i64],
    "capi_block_to_cyclic_tri": [_vp, _vp, _i64, _i64],
    "capi_cyclic_to_block_tri": [_vp, _vp, _i64, _i64],
    "capi_cyclic_to_local": [_vp, _vp, _i64, _i64, _i64, _i64],
    "capi_distribute_symmetric": [_vp] + [_i64] * 9 + [_int],
    "capi_distribute_random": [_vp] + [_i64] * 9,
    "capi_distribute_identity": [_vp] + [_i64] * 8 + [_dbl],
    "capi_diff_norms": [_int, _i64, _i64, _vp, _i64, _vp, _i64, C.POINTER(_dbl)],
    "capi_mfma_f64_peak": [_int, C.POINTER(_dbl)],
    "capi_prof_enable": [_int],
    "capi_prof_collect": [_int, C.POINTER(_i64), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl)],
    "capi_prof_collect_intervals": [_int, C.POINTER(_i64), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl)],
    "capi_stream_select": [_int],
    "capi_event_record": [_int],
    "capi_event_wait": [_int],
    "capi_timer_start": [],
    "capi_timer_stop_ms": [C.POINTER(C.c_float)],
    "capi_sync": [],
    "capi_dgeqrf": [_i64, _i64, _vp, _i64, _vp],
    "capi_dorgqr": [_i64, _i64, _i64, _vp, _i64, _vp],
    "capi_reserve_workspace": [C.c_size_t],
    "capi_trim_workspaces": [],
    "capi_set_launch_rounds": [_int, C.POINTER(_int)],
    "capi_reserve_cus": [_int],
    "capi_memset_async": [_vp, _int, C.c_size_t],
    "capi_memcpy_h2d": [_vp, _vp, C.c_size_t],
    "capi_memcpy_d2h": [_vp, _vp, C.c_size_t],
    "capi_memcpy_d2d_async": [_vp, _vp, C.c_size_t],
    "capi_malloc": [C.POINTER(_vp), C.c_size_t],
    "capi_free": [_vp],
}
_COMM_SIGS = {
    "capi_comm_split": [_vp, _int, _int, C.POINTER(_vp)],
    "capi_comm_rank": [_vp, C.POINTER(_int)],
    "capi_comm_size": [_vp, C.POINTER(_int)],
    "capi_comm_destroy": [_vp],
    "capi_bcast": [_vp, _vp, _i64, _int],
    "capi_allreduce_sum": [_vp, _vp, _i64],
    "capi_reduce_sum": [_vp, _vp, _vp, _i64, _int],
    "capi_allgather": [_vp, _vp, _vp, _i64],
    "capi_sendrecv_replace": [_vp, _vp, _i64, _int, _vp],
    "capi_gather": [_vp, _vp, _vp, _i64, _int],
    "capi_scatter": [_vp, _vp, _vp, _i64, _int],
    "capi_comm_query": [_vp, C.POINTER(_int), C.POINTER(_int)],
    "capi_pairs_transfer": [_vp, C.POINTER(_int), _vp, _vp, _i64, _vp],
}

_lib = None


class CapiError(RuntimeError):
    pass


def load():
    """Load libcapital_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CapiError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C capital_amd/csrc).  capital_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, sig in _SIGS.items():
        f = getattr(L, name)
        f.argtypes = [_vp] + sig
        f.restype = _int
    for name, sig in _COMM_SIGS.items():
        f = getattr(L, name)
        f.argtypes = sig
        f.restype = _int
    L.capi_create.argtypes = [C.POINTER(_vp), _int]
    L.capi_create_on_stream.argtypes = [C.POINTER(_vp), _int, _vp]
    L.capi_destroy.argtypes = [_vp]
    L.capi_get_stream.argtypes = [_vp]
    L.capi_get_stream.restype = _vp
    L.capi_last_error.argtypes = [_vp]
    L.capi_last_error.restype = C.c_char_p
    L.capi_comm_load_rccl.argtypes = [C.c_char_p]
    L.capi_comm_unique_id.argtypes = [_vp]
    L.capi_comm_init_rank.argtypes = [C.POINTER(_vp), _vp, _int, _vp, _int]
    L.capi_version.restype = _int
    L.capi_pairs_scratch_count.argtypes = [_int, _i64]
    L.capi_pairs_scratch_count.restype = _i64
    L.capi_device_count.restype = _int
    _lib = L
    return L


def declared_symbols():
    """Every function name include/capital_hip.h declares (used by the CPU export test)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "capital_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"^(?:int|int64_t|void\*|const char\*)\s+(capi_[a-z0-9_]+)\s*\(", txt, re.M)))


def ptr(t):
    """Device pointer of a torch tensor (or pass through ints / None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


class Handle:
    """capi_handle_t bound to a device; by default it borrows torch's current stream so that
    torch allocations / copies and capi kernels are ordered on one stream."""

    def __init__(self, device=0, own_stream=False):
        L = load()
        if not torch.cuda.is_available():
            raise CapiError("capital_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.L = L
        self.device = device
        h = _vp()
        torch.cuda.set_device(device)
        if own_stream:
            rc = L.capi_create(C.byref(h), device)
        else:
            rc = L.capi_create_on_stream(C.byref(h), device, _vp(torch.cuda.current_stream(device).cuda_stream))
        if rc != 0:
            raise CapiError(f"capi_create failed: {rc}")
        self.h = h

    def call(self, name, *args):
        rc = getattr(self.L, name)(self.h, *args)
        if rc != 0:
            raise CapiError(f"{name} -> {rc}: {self.L.capi_last_error(self.h).decode()}")

    def info(self):
        v = _int(0)
        self.call("capi_get_info", C.byref(v))
        return v.value

    def sync(self):
        self.call("capi_sync")

    def close(self):
        if self.h:
            self.L.capi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- column-major helpers: a column-major m x n matrix with ld = m is a row-major (n, m) torch tensor ----
def to_device(a, device=0):
    """numpy (m, n) -> device tensor holding it column-major (shape (n, m), contiguous)."""
    a = np.asarray(a, dtype=np.float64)
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(f"cuda:{device}")


def to_host(t):
    """inverse of to_device: (n, m) device tensor -> numpy (m, n) Fortran-ordered."""
    return np.asfortranarray(t.cpu().numpy().T)


def empty(m, n, device=0):
    return torch.empty((n, m), dtype=torch.float64, device=f"cuda:{device}")


def zeros(m, n, device=0):
    return torch.zeros((n, m), dtype=torch.float64, device=f"cuda:{device}")
