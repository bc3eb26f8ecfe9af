"""Regenerates tests/golden/*.npz.  The reference ships no golden vectors and cannot be built in this image
(mkl.h is absent), so these fixtures are produced from (a) glibc srand48/drand48 -- the functions the reference's
generators call -- and (b) LAPACK as installed here (scipy), i.e. independently of oracle/ and of the HIP code.
They pin BOTH: tests/test_golden.py checks the oracle against them on CPU and the GPU path against them on MI355X.

    python tests/golden/make_golden.py
"""
import ctypes
import ctypes.util
import os

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
libc = ctypes.CDLL(ctypes.util.find_library("c"))
libc.srand48.argtypes = [ctypes.c_long]
libc.drand48.restype = ctypes.c_double


def spd(n):
    """distribute_symmetric(..., diagonallyDominant=true) on a 1x1 grid, structure.hpp:68-103"""
    G = np.zeros((n, n))
    for gx in range(n):
        for gy in range(n):
            libc.srand48(gx + n * gy if gx > gy else gy + n * gx)
            G[gy, gx] = libc.drand48() + (n if gx == gy else 0)
    return G


def tall(m, n, key=0):
    """distribute_random on a 1-rank grid, structure.hpp:105-129 (column-major draw order)"""
    libc.srand48(key)
    A = np.zeros((m, n))
    for i in range(n):
        for j in range(m):
            A[j, i] = libc.drand48()
    return A


def main():
    out = {}
    for n in (64, 96):
        A = spd(n)
        R = sl.cholesky(A, lower=False)
        Ri = sl.solve_triangular(R, np.eye(n), lower=False)
        out[f"spd_{n}"] = A
        out[f"R_{n}"] = np.triu(R)
        out[f"Rinv_{n}"] = np.triu(Ri)
    m, n = 512, 24
    A = tall(m, n)
    Q, R = np.linalg.qr(A)
    s = np.sign(np.diag(R))
    out["tall_512x24"] = A
    out["Q_512x24"] = Q * s
    out["Rq_512x24"] = (R.T * s).T
    np.savez_compressed(os.path.join(HERE, "capital_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "capital_golden.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
