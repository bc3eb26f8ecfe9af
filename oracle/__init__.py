"""ctypes binding of the CPU ORACLE (oracle/capital_oracle.c).

TEST INFRASTRUCTURE ONLY.  Allowed importers: tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product package (capital_amd) never
imports this module and has no CPU fallback.

Matrices are numpy float64 arrays in Fortran (column-major) order.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcapital_oracle.so")

NOTRANS, TRANS = 0, 1
LEFT, RIGHT = 0, 1
LOWER, UPPER = 0, 1
NONUNIT, UNIT = 0, 1


def build(force=False):
    """Compile oracle/capital_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "capital_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libcapital_oracle.so"])
    return _LIB_PATH


_lib = None
_i64, _dbl, _int = C.c_int64, C.c_double, C.c_int
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_dgemm.argtypes = [_int, _int, _i64, _i64, _i64, _dbl, _dp, _i64, _dp, _i64, _dbl, _dp, _i64]
        L.orc_dtrmm.argtypes = [_int, _int, _int, _int, _i64, _i64, _dbl, _dp, _i64, _dp, _i64]
        L.orc_dtrsm.argtypes = [_int, _int, _int, _int, _i64, _i64, _dbl, _dp, _i64, _dp, _i64]
        L.orc_dsyrk.argtypes = [_int, _int, _i64, _i64, _dbl, _dp, _i64, _dbl, _dp, _i64]
        L.orc_dpotrf.argtypes = [_int, _i64, _dp, _i64]
        L.orc_dpotrf.restype = _int
        L.orc_dtrtri.argtypes = [_int, _int, _i64, _dp, _i64]
        L.orc_dtrtri.restype = _int
        L.orc_dgeqrf.argtypes = [_i64, _i64, _dp, _i64, _dp]
        L.orc_dgeqrf.restype = _int
        L.orc_dorgqr.argtypes = [_i64, _i64, _i64, _dp, _i64, _dp]
        L.orc_dorgqr.restype = _int
        for nm in ("orc_distribute_symmetric",):
            getattr(L, nm).argtypes = [_dp] + [_i64] * 9 + [_int]
        L.orc_distribute_random.argtypes = [_dp] + [_i64] * 9
        L.orc_distribute_identity.argtypes = [_dp] + [_i64] * 8 + [_dbl]
        L.orc_drand48_after_seed.argtypes = [_i64]
        L.orc_drand48_after_seed.restype = _dbl
        L.orc_drand48_stream.argtypes = [_i64, _i64, _dp]
        L.orc_offset.argtypes = [_int, _i64, _i64, _i64, _i64]
        L.orc_offset.restype = _i64
        L.orc_serialize.argtypes = [_int, _int, _dp, _i64, _i64, _dp, _i64, _i64] + [_i64] * 8
        L.orc_block_to_cyclic_rect.argtypes = [_dp, _dp, _i64, _i64, _i64]
        L.orc_cyclic_to_block_rect.argtypes = [_dp, _dp, _i64, _i64, _i64]
        L.orc_block_to_cyclic_triangle.argtypes = [_dp, _dp, _i64, _i64, _i64, _i64]
        L.orc_cyclic_to_block_triangle.argtypes = [_dp, _dp, _i64, _i64, _i64, _i64]
        L.orc_cyclic_to_local.argtypes = [_dp, _dp, _i64, _i64, _i64, _i64]
        L.orc_cyclic_extract.argtypes = [_dp, _i64, _i64, _i64, _dp, _i64, _i64, _i64, _i64]
        L.orc_cyclic_insert.argtypes = [_dp, _i64, _i64, _i64, _dp, _i64, _i64, _i64, _i64]
        L.orc_cholinv_factor.argtypes = [_dp, _i64, _int, _int, _int, _int, _int, _dp, _dp]
        L.orc_cholinv_factor.restype = _int
        L.orc_cholinv_bc_dimension.argtypes = [_i64, _int, _int, _int]
        L.orc_cholinv_bc_dimension.restype = _i64
        L.orc_cacqr_factor_1d.argtypes = [_dp, _i64, _i64, _int, _int, _dp]
        L.orc_cacqr_factor_1d.restype = _int
        L.orc_cholesky_residual.argtypes = [_dp, _dp, _i64]
        L.orc_cholesky_residual.restype = _dbl
        L.orc_qr_residual.argtypes = [_dp, _dp, _dp, _i64, _i64]
        L.orc_qr_residual.restype = _dbl
        L.orc_qr_orthogonality.argtypes = [_dp, _i64, _i64]
        L.orc_qr_orthogonality.restype = _dbl
        L.orc_set_threads.argtypes = [_int]
        L.orc_get_threads.restype = _int
        # OpenMP's default is one thread per hardware thread the affinity mask lists -- 128 on a one-GPU box whose cgroup share is 16 cores:
        # the checker then spins instead of computing (a 2-rank parity run of nine n = 4096 cases went from ~1 to > 8 minutes when nothing
        # else in the process had tamed the runtime first).  Cap it at the share unless the caller chose (OMP_NUM_THREADS).
        if not os.environ.get("OMP_NUM_THREADS"):
            L.orc_set_threads(_cpu_share())
        _lib = L
    return _lib


def _cpu_share():
    """cores this process may really use: cgroup CPU quota, else the affinity mask; at most 16 (see oracle/host_baseline.py)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def _p(a):
    assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"], "oracle wants float64 column-major arrays"
    return a.ctypes.data_as(_dp)


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def set_threads(n):
    lib().orc_set_threads(int(n))


def get_threads():
    return lib().orc_get_threads()


# ---- host-BLAS backend (capital_oracle.c: orc_host_blas_bind) -------------------------------
def _host_blas_candidates():
    """(path, prefix, suffix, ilp64, family) in the order SURVEY.md 8(d)(i) prescribes: MKL, OpenBLAS, then the OpenBLAS
    builds bundled with scipy / numpy."""
    import glob
    out = []
    for d in ("", "/opt/conda/lib/", "/usr/lib/x86_64-linux-gnu/", "/opt/intel/oneapi/mkl/latest/lib/"):
        for nm in ("libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so"):
            out.append((d + nm, "", "", 0, "mkl"))
    for d in ("", "/usr/lib/x86_64-linux-gnu/", "/opt/conda/lib/"):
        for nm in ("libopenblas.so.0", "libopenblas.so"):
            out.append((d + nm, "", "", 0, "openblas"))
    for pkg, ilp in (("scipy", 0), ("numpy", 1)):
        try:
            mod = __import__(pkg)
        except ImportError:
            continue
        base = os.path.join(os.path.dirname(os.path.dirname(mod.__file__)), pkg + ".libs")
        for f in sorted(glob.glob(os.path.join(base, "libscipy_openblas*.so"))):
            is64 = "openblas64_" in os.path.basename(f)
            out.append((f, "scipy_", "64_" if is64 else "", 1 if is64 else 0, "openblas"))
    return out


def bind_host_blas(threads=None):
    """Route dgemm/dtrmm/dsyrk/dpotrf/dtrtri of this oracle to the first host BLAS/LAPACK library found.
    Returns {"library", "path", "threads"} or None when the box has none (the oracle's own kernels stay in use)."""
    L = lib()
    L.orc_host_blas_bind.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, _int]
    L.orc_host_blas_bind.restype = _int
    for path, pre, suf, ilp, fam in _host_blas_candidates():
        if os.path.sep in path and not os.path.exists(path):
            continue
        if fam == "mkl":
            # this library is built with gcc's OpenMP: MKL's default Intel threading layer would bring a second OpenMP runtime
            # into the process (measured here: n = 4096 schedule 1.66 s with both runtimes, 0.29 s on the GNU layer)
            os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        if L.orc_host_blas_bind(path.encode(), pre.encode(), suf.encode(), ilp) != 0:
            continue
        H = C.CDLL(path)
        nthr = int(threads or os.cpu_count() or 1)
        name = fam
        try:
            if fam == "mkl":
                H.MKL_Set_Num_Threads(C.c_int(nthr))
                H.MKL_Get_Max_Threads.restype = C.c_int
                nthr = int(H.MKL_Get_Max_Threads())
                buf = C.create_string_buffer(256)
                H.MKL_Get_Version_String(buf, C.c_int(256))
                name = buf.value.decode(errors="replace").strip() or "Intel MKL"
            else:
                getattr(H, pre + "openblas_set_num_threads" + suf)(C.c_int(nthr))
                cfg = getattr(H, pre + "openblas_get_config" + suf)
                cfg.restype = C.c_char_p
                name = cfg().decode(errors="replace").strip()
        except (AttributeError, OSError):
            pass
        return {"library": name, "path": path, "threads": nthr, "ilp64": bool(ilp)}
    return None


def host_blas_enable(on):
    lib().orc_host_blas_enable(int(bool(on)))


def host_blas_active():
    return bool(lib().orc_host_blas_active())


# ---- K1..K9 -------------------------------------------------------------------------------
def dgemm(transA, transB, alpha, A, B, beta, Cm):
    """C <- alpha op(A) op(B) + beta C, in place on Cm (column-major)."""
    m, n = Cm.shape
    k = A.shape[0] if transA else A.shape[1]
    lib().orc_dgemm(transA, transB, m, n, k, alpha, _p(A), A.shape[0], _p(B), B.shape[0], beta, _p(Cm), Cm.shape[0])
    return Cm


def dtrmm(side, uplo, trans, diag, alpha, T, B):
    m, n = B.shape
    lib().orc_dtrmm(side, uplo, trans, diag, m, n, alpha, _p(T), T.shape[0], _p(B), B.shape[0])
    return B


def dtrsm(side, uplo, trans, diag, alpha, T, B):
    m, n = B.shape
    lib().orc_dtrsm(side, uplo, trans, diag, m, n, alpha, _p(T), T.shape[0], _p(B), B.shape[0])
    return B


def dsyrk(uplo, trans, alpha, A, beta, Cm):
    n = Cm.shape[0]
    k = A.shape[0] if trans else A.shape[1]
    lib().orc_dsyrk(uplo, trans, n, k, alpha, _p(A), A.shape[0], beta, _p(Cm), Cm.shape[0])
    return Cm


def dpotrf(uplo, A, n=None):
    n = A.shape[0] if n is None else n
    return lib().orc_dpotrf(uplo, n, _p(A), A.shape[0])


def dtrtri(uplo, diag, A, n=None):
    n = A.shape[0] if n is None else n
    return lib().orc_dtrtri(uplo, diag, n, _p(A), A.shape[0])


def dgeqrf(A):
    """In place on a Fortran-ordered (m, n) array: R above, reflectors below the diagonal; returns tau (min(m, n))."""
    m, n = A.shape
    tau = np.zeros(min(m, n))
    lib().orc_dgeqrf(m, n, _p(A), A.shape[0], _p(tau))
    return tau


def dorgqr(A, tau, k=None):
    """In place: the first n columns of Q = H_1 ... H_k from dgeqrf's output."""
    m, n = A.shape
    rc = lib().orc_dorgqr(m, n, len(tau) if k is None else k, _p(A), A.shape[0], _p(tau))
    assert rc == 0
    return A


# ---- generators -----------------------------------------------------------------------------
def local_dims(gdimX, gdimY, PX, PY):
    """matrix.hpp:8-11: local extents (columns, rows) with at most one padded row/column."""
    return gdimX // PX + (1 if gdimX % PX else 0), gdimY // PY + (1 if gdimY % PY else 0)


def distribute_symmetric(gdimX, gdimY, px, py, PX, PY, key=0, diag_dominant=True):
    dx, dy = local_dims(gdimX, gdimY, PX, PY)
    a = np.zeros((dy, dx), order="F")
    lib().orc_distribute_symmetric(_p(a), dx, dy, gdimX, gdimY, px, py, PX, PY, key, int(diag_dominant))
    return a


def distribute_random(gdimX, gdimY, px, py, PX, PY, key=0):
    dx, dy = local_dims(gdimX, gdimY, PX, PY)
    a = np.zeros((dy, dx), order="F")
    lib().orc_distribute_random(_p(a), dx, dy, gdimX, gdimY, px, py, PX, PY, key)
    return a


def distribute_identity(gdimX, gdimY, px, py, PX, PY, val=1.0):
    dx, dy = local_dims(gdimX, gdimY, PX, PY)
    a = np.zeros((dy, dx), order="F")
    lib().orc_distribute_identity(_p(a), dx, dy, gdimX, gdimY, px, py, PX, PY, val)
    return a


def drand48_after_seed(seed):
    return lib().orc_drand48_after_seed(int(seed))


def drand48_stream(seed, count):
    out = np.zeros(count)
    lib().orc_drand48_stream(int(seed), count, out.ctypes.data_as(_dp))
    return out


def cyclic_extract(G, px, py, PX, PY):
    grows, gcols = G.shape
    dx, dy = local_dims(gcols, grows, PX, PY)
    loc = np.zeros((dy, dx), order="F")
    lib().orc_cyclic_extract(_p(G), gcols, grows, G.shape[0], _p(loc), px, py, PX, PY)
    return loc


def cyclic_insert(G, loc, px, py, PX, PY):
    grows, gcols = G.shape
    lib().orc_cyclic_insert(_p(G), gcols, grows, G.shape[0], _p(loc), px, py, PX, PY)
    return G


# ---- schedules ------------------------------------------------------------------------------
def cholinv_factor(A, complete_inv=0, split=1, bc_mult_dim=0, c=1, d=1, out=None):
    """Returns (R, Rinv, info); A is n x n symmetric (only its upper triangle is read).  `out` = (R, Rinv) reuses two
    column-major n x n arrays (both are overwritten entirely)."""
    A = _f(A)
    n = A.shape[0]
    if out is not None:
        R, Rinv = out
        assert R.shape == (n, n) and Rinv.shape == (n, n) and R.flags.f_contiguous and Rinv.flags.f_contiguous
    else:
        R = np.zeros((n, n), order="F")
        Rinv = np.zeros((n, n), order="F")
    info = lib().orc_cholinv_factor(_p(A), n, int(complete_inv), int(split), int(bc_mult_dim), c, d, _p(R), _p(Rinv))
    return R, Rinv, info


def cholinv_bc_dimension(n_local, c, d, bc_mult_dim):
    return lib().orc_cholinv_bc_dimension(n_local, c, d, bc_mult_dim)


def cacqr_factor_1d(A, P=1, num_iter=2):
    """Returns (Q, R, info) of CholeskyQR (num_iter=1) / CholeskyQR2 (num_iter=2) on P simulated ranks."""
    Q = _f(A).copy(order="F")
    m, n = Q.shape
    R = np.zeros((n, n), order="F")
    info = lib().orc_cacqr_factor_1d(_p(Q), m, n, P, num_iter, _p(R))
    return Q, R, info


# ---- validators -----------------------------------------------------------------------------
def cholesky_residual(A, R):
    A, R = _f(A), _f(R)
    return lib().orc_cholesky_residual(_p(A), _p(R), A.shape[0])


def qr_residual(A, Q, R):
    A, Q, R = _f(A), _f(Q), _f(R)
    return lib().orc_qr_residual(_p(A), _p(Q), _p(R), A.shape[0], A.shape[1])


def qr_orthogonality(Q):
    Q = _f(Q)
    return lib().orc_qr_orthogonality(_p(Q), Q.shape[0], Q.shape[1])
