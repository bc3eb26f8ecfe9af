"""ctypes binding of libcapital_driver.so (capital_amd/drivers/capital_driver.cpp): the host-side C++ layer
(cholesky::cholinv<...>::factor, qr::cacqr<...>::factor, validators) behind plain C entry points.

One process per GPU.  `init()` binds the C++ layer to this process's device/stream and, for world_size > 1, builds the
RCCL world communicator from a unique id that rank 0 creates and torch.distributed ships to the other ranks.
No CPU fallback: everything here ends in libcapital_hip.so kernels.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
DRV_PATH = os.environ.get("CAPITAL_DRIVER_LIB", os.path.join(_HERE, "libcapital_driver.so"))   # override: A/B builds of the same ABI
_i64, _dbl, _int, _vp = C.c_int64, C.c_double, C.c_int, C.c_void_p
_dp = C.POINTER(C.c_double)
_drv = None
_state = {"init": False, "rank": 0, "size": 1}


class DriverError(RuntimeError):
    pass


def load():
    global _drv
    if _drv is not None:
        return _drv
    capi.load()  # libcapital_hip.so first (RTLD_GLOBAL), so the driver resolves against it
    if not os.path.exists(DRV_PATH):
        raise DriverError(f"{DRV_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    D = C.CDLL(DRV_PATH, mode=C.RTLD_GLOBAL)
    _drv = bind(D)
    return D


def bind(D):
    """Declare the driver's C signatures on a loaded library object."""
    D.capital_drv_last_error.restype = C.c_char_p
    D.capital_drv_init.argtypes = [_int, _int, _int, _vp, _vp]
    D.capital_drv_handle.restype = _vp
    D.capital_cholinv_create.argtypes = [_i64] + [_int] * 8
    D.capital_cholinv_create.restype = _vp
    for f in ("generate", "factor", "destroy"):
        getattr(D, f"capital_cholinv_{f}").argtypes = [_vp]
        getattr(D, f"capital_cacqr_{f}").argtypes = [_vp]
    D.capital_cholinv_set_A.argtypes = [_vp, _dp]
    D.capital_cholinv_residual.argtypes = [_vp, _dp]
    D.capital_cholinv_get.argtypes = [_vp, _int, _dp]
    D.capital_cholinv_dims.argtypes = [_vp, C.POINTER(_i64)] + [C.POINTER(_int)] * 5
    D.capital_cholinv_stats.argtypes = [_vp] + [C.POINTER(_i64)] * 3
    D.capital_cholinv_set_trsm_mode.argtypes = [_vp, _int]
    D.capital_cacqr_create.argtypes = [_i64, _i64] + [_int] * 8
    D.capital_cacqr_create.restype = _vp
    D.capital_cacqr_set_A.argtypes = [_vp, _dp]
    D.capital_cacqr_residual.argtypes = [_vp, _dp]
    D.capital_cacqr_orthogonality.argtypes = [_vp, _dp]
    D.capital_cacqr_get.argtypes = [_vp, _int, _dp]
    D.capital_cacqr_dims.argtypes = [_vp, C.POINTER(_i64), C.POINTER(_i64)]
    D.capital_cacqr_get_rows.argtypes = [_vp, _int, _i64, _i64, _dp]
    return D


def _ck(rc, what):
    if rc != 0:
        raise DriverError(f"{what}: {load().capital_drv_last_error().decode()}")


def init(device=0, rank=0, size=1, unique_id=None, use_torch_stream=True):
    """Bind the C++ layer to `device`.  With size > 1 `unique_id` is the 128-byte RCCL id from rank 0."""
    D = load()
    if not torch.cuda.is_available():
        raise DriverError("capital_amd needs an AMD GPU; there is no CPU fallback")
    torch.cuda.set_device(device)
    stream = _vp(torch.cuda.current_stream(device).cuda_stream) if use_torch_stream else None
    if use_torch_stream and not stream.value:
        # torch's default stream is the NULL stream; give the layer its own stream instead and sync explicitly
        stream = None
    uid = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
    _ck(D.capital_drv_init(device, rank, size, uid, stream), "capital_drv_init")
    _state.update(init=True, rank=rank, size=size)


def init_distributed(device):
    """torch.distributed must be initialised; ships the RCCL unique id from rank 0 and builds the world communicator."""
    import torch.distributed as dist
    rank, size = dist.get_rank(), dist.get_world_size()
    uid = None
    if size > 1 or os.environ.get("CAPI_RCCL_FORCE"):      # (CAPI_RCCL_FORCE: even a 1-rank communicator goes through RCCL)
        L = capi.load()
        torch_rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        named = os.environ.get("CAPI_RCCL_LIB")          # an RCCL build of the caller's choosing (same ABI), e.g. a debug build
        if named:
            if L.capi_comm_load_rccl(named.encode()) != 0:
                raise DriverError(f"CAPI_RCCL_LIB={named} cannot be loaded as an RCCL library")
        elif os.path.exists(torch_rccl):
            L.capi_comm_load_rccl(torch_rccl.encode())   # one RCCL per process: the copy torch maps
        buf = (C.c_char * 128)()
        if rank == 0:
            rc = L.capi_comm_unique_id(buf)
            if rc != 0:
                raise DriverError(f"capi_comm_unique_id -> {rc}")
        obj = [bytes(buf)]
        dist.broadcast_object_list(obj, src=0)
        uid = obj[0]
    init(device, rank, size, uid, use_torch_stream=False)


def finalize():
    if _state["init"]:
        load().capital_drv_finalize()
        _state["init"] = False


def sync():
    _ck(load().capital_drv_sync(), "sync")


def handle_ptr():
    return load().capital_drv_handle()


def world_query():
    """(rank, size) of the world communicator as RCCL reports them."""
    r, s = _int(), _int()
    _ck(load().capital_drv_world_query(C.byref(r), C.byref(s)), "world_query")
    return r.value, s.value


class Cholinv:
    """cholesky::cholinv<SP,SaveIntermediates,BP>::factor on this process's block of an n x n SPD matrix.
    Arguments in the order of the reference bench (bench/cholesky/cholinv.cpp:15-22)."""

    def __init__(self, n, c=1, complete_inv=0, split=1, bc_mult=0, layout=0, num_chunks=0, serialize=True, bc_policy=2, trsm_mode=False,
                 flush_intermediates=False):
        self.D = load()
        self.p = self.D.capital_cholinv_create(n, c, layout, num_chunks, int(complete_inv), split, bc_mult,
                                               int(bool(serialize)) + (2 if flush_intermediates else 0), bc_policy)
        if not self.p:
            raise DriverError("capital_cholinv_create: " + self.D.capital_drv_last_error().decode())
        if trsm_mode:      # info::solve_with_trsm: potrf + block TRSM + SYRK, no inverse formed (not the reference's schedule)
            _ck(self.D.capital_cholinv_set_trsm_mode(self.p, 1), "set_trsm_mode")
        nloc, x, y, z, d, cc = _i64(), _int(), _int(), _int(), _int(), _int()
        _ck(self.D.capital_cholinv_dims(self.p, C.byref(nloc), C.byref(x), C.byref(y), C.byref(z), C.byref(d), C.byref(cc)), "dims")
        self.n, self.n_loc, self.x, self.y, self.z, self.d, self.c = n, nloc.value, x.value, y.value, z.value, d.value, cc.value

    def generate(self):
        _ck(self.D.capital_cholinv_generate(self.p), "generate")

    def set_A(self, A_local):
        a = np.asfortranarray(A_local, dtype=np.float64)
        assert a.shape == (self.n_loc, self.n_loc)
        _ck(self.D.capital_cholinv_set_A(self.p, a.ctypes.data_as(_dp)), "set_A")

    def factor(self):
        _ck(self.D.capital_cholinv_factor(self.p), "factor")

    def residual(self):
        v = _dbl()
        _ck(self.D.capital_cholinv_residual(self.p, C.byref(v)), "residual")
        return v.value

    def _get(self, which):
        out = np.zeros((self.n_loc, self.n_loc), order="F")
        _ck(self.D.capital_cholinv_get(self.p, which, out.ctypes.data_as(_dp)), "get")
        return out

    def A(self):
        return self._get(0)

    def R(self):
        return self._get(1)

    def Rinv(self):
        return self._get(2)

    def stats(self):
        a, b, c_ = _i64(), _i64(), _i64()
        _ck(self.D.capital_cholinv_stats(self.p, C.byref(a), C.byref(b), C.byref(c_)), "stats")
        return {"base_cases": a.value, "levels": b.value, "bc_dimension": c_.value}

    def close(self):
        if self.p:
            self.D.capital_cholinv_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cacqr:
    """qr::cacqr<SP,SaveIntermediates>::factor on this process's row-cyclic block of an m x n matrix
    (bench/qr/cacqr.cpp:14-25: variant 1 = CholeskyQR, 2 = CholeskyQR2)."""

    def __init__(self, m, n, c=1, variant=2, complete_inv=0, split=1, bc_mult=0, layout=0, num_chunks=0, serialize=True):
        self.D = load()
        self.p = self.D.capital_cacqr_create(m, n, c, variant, layout, num_chunks, int(complete_inv), split, bc_mult, int(serialize))
        if not self.p:
            raise DriverError("capital_cacqr_create: " + self.D.capital_drv_last_error().decode())
        ml, nn = _i64(), _i64()
        _ck(self.D.capital_cacqr_dims(self.p, C.byref(ml), C.byref(nn)), "dims")
        self.m, self.n, self.m_loc = m, nn.value, ml.value

    def generate(self):
        _ck(self.D.capital_cacqr_generate(self.p), "generate")

    def set_A(self, A_local):
        a = np.asfortranarray(A_local, dtype=np.float64)
        assert a.shape == (self.m_loc, self.n)
        _ck(self.D.capital_cacqr_set_A(self.p, a.ctypes.data_as(_dp)), "set_A")

    def factor(self):
        _ck(self.D.capital_cacqr_factor(self.p), "factor")

    def residual(self):
        v = _dbl()
        _ck(self.D.capital_cacqr_residual(self.p, C.byref(v)), "residual")
        return v.value

    def orthogonality(self):
        v = _dbl()
        _ck(self.D.capital_cacqr_orthogonality(self.p, C.byref(v)), "orthogonality")
        return v.value

    def _get(self, which, shape):
        out = np.zeros(shape, order="F")
        _ck(self.D.capital_cacqr_get(self.p, which, out.ctypes.data_as(_dp)), "get")
        return out

    def A(self):
        return self._get(0, (self.m_loc, self.n))

    def Q(self):
        return self._get(1, (self.m_loc, self.n))

    def R(self):
        return self._get(2, (self.n, self.n))

    def gram_of_Q(self):
        """Q^T Q summed over the ranks (n x n): what the orthogonality validator measures against I."""
        return self._get(3, (self.n, self.n))

    def rows(self, which, row0, nrows):
        """Local rows [row0, row0 + nrows) of A (which = 'A') or Q ('Q') -- a bounded window of a panel too large to fetch whole."""
        out = np.zeros((nrows, self.n), order="F")
        _ck(self.D.capital_cacqr_get_rows(self.p, {"A": 0, "Q": 1}[which], row0, nrows, out.ctypes.data_as(_dp)), "get_rows")
        return out

    def close(self):
        if self.p:
            self.D.capital_cacqr_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
