"""Helper of tests/test_oracle_blas.py::test_against_mkl_entry_points (run in its own interpreter)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

mkl = None
for p in ("/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so"):
    if os.path.exists(p):
        mkl = C.CDLL(p)
        break
assert mkl is not None
RNG = np.random.default_rng(42)


def rnd(m, n):
    return np.asfortranarray(RNG.uniform(-1, 1, (m, n)))


def close(a, b, k=1):
    assert np.abs(a - b).max() <= 2e-14 * max(k, 8) * max(1.0, np.abs(b).max()), np.abs(a - b).max()


dp = C.POINTER(C.c_double)
COL, NOT, TR, UP, NONU, LEFT = 102, 111, 112, 121, 131, 141
mkl.cblas_dgemm.argtypes = [C.c_int] * 6 + [C.c_double, dp, C.c_int, dp, C.c_int, C.c_double, dp, C.c_int]
mkl.cblas_dtrmm.argtypes = [C.c_int] * 7 + [C.c_double, dp, C.c_int, dp, C.c_int]
mkl.cblas_dsyrk.argtypes = [C.c_int] * 5 + [C.c_double, dp, C.c_int, C.c_double, dp, C.c_int]
mkl.LAPACKE_dpotrf.argtypes = [C.c_int, C.c_char, C.c_int, dp, C.c_int]
mkl.LAPACKE_dtrtri.argtypes = [C.c_int, C.c_char, C.c_char, C.c_int, dp, C.c_int]
m, n, k = 120, 90, 75
A, B, Cm = rnd(k, m), rnd(k, n), rnd(m, n)
ref = Cm.copy(order="F")
mkl.cblas_dgemm(COL, TR, NOT, m, n, k, -1.0, A.ctypes.data_as(dp), k, B.ctypes.data_as(dp), k, 1.0, ref.ctypes.data_as(dp), m)
close(oracle.dgemm(1, 0, -1.0, A, B, 1.0, Cm.copy(order="F")), ref, k)
T, Bm = np.asfortranarray(rnd(m, m) + 4 * np.eye(m)), rnd(m, n)
ref = Bm.copy(order="F")
mkl.cblas_dtrmm(COL, LEFT, UP, TR, NONU, m, n, 1.0, T.ctypes.data_as(dp), m, ref.ctypes.data_as(dp), m)
close(oracle.dtrmm(0, 1, 1, 0, 1.0, T, Bm.copy(order="F")), ref, m)
S = oracle.distribute_symmetric(200, 200, 0, 0, 1, 1)
ref = S.copy(order="F")
assert mkl.LAPACKE_dpotrf(COL, b"U", 200, ref.ctypes.data_as(dp), 200) == 0
got = S.copy(order="F")
assert oracle.dpotrf(1, got) == 0
close(np.triu(got), np.triu(ref), 200)
assert mkl.LAPACKE_dtrtri(COL, b"U", b"N", 200, ref.ctypes.data_as(dp), 200) == 0
assert oracle.dtrtri(1, 0, got) == 0
close(np.triu(got), np.triu(ref), 200)
G = np.zeros((n, n), order="F")
mkl.cblas_dsyrk(COL, UP, TR, n, k, 1.0, B.ctypes.data_as(dp), k, 0.0, G.ctypes.data_as(dp), n)
close(np.triu(oracle.dsyrk(1, 1, 1.0, B, 0.0, np.zeros((n, n), order="F"))), np.triu(G), k)
print("MKL-OK")
