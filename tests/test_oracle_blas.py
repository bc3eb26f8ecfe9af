"""CPU: pin the oracle's restatement of the seven MKL entry points the reference calls against
the BLAS/LAPACK this image does have: scipy's (OpenBLAS) always, and libmkl_rt.so -- the reference's own third-party
dependency -- when it is installed.  fp64 tolerance 1e-13 * k relative (summation order only)."""
import ctypes as C
import itertools
import os

import numpy as np
import pytest
import scipy.linalg.blas as sb
import scipy.linalg.lapack as sl

RNG = np.random.default_rng(42)


def rnd(m, n):
    return np.asfortranarray(RNG.uniform(-1, 1, (m, n)))


def close(a, b, k=1):
    assert np.abs(a - b).max() <= 2e-14 * max(k, 8) * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("ta,tb", list(itertools.product((0, 1), (0, 1))))
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (7, 5, 3), (97, 131, 260), (300, 64, 1000)])
def test_dgemm(oracle, ta, tb, m, n, k):
    A = rnd(k, m) if ta else rnd(m, k)
    B = rnd(n, k) if tb else rnd(k, n)
    Cm = rnd(m, n)
    ref = sb.dgemm(0.7, A, B, beta=-0.3, c=Cm, trans_a=ta, trans_b=tb)
    got = oracle.dgemm(ta, tb, 0.7, A, B, -0.3, Cm.copy(order="F"))
    close(got, ref, k)


@pytest.mark.parametrize("side,uplo,trans,diag", list(itertools.product((0, 1), (0, 1), (0, 1), (0, 1))))
def test_dtrmm_dtrsm(oracle, side, uplo, trans, diag):
    m, n = 150, 133
    nt = m if side == 0 else n
    T = np.asfortranarray(rnd(nt, nt) * 0.1 + 3 * np.eye(nt))
    B = rnd(m, n)
    ref = sb.dtrmm(1.25, T, B, side=side, lower=1 - uplo, trans_a=trans, diag=diag)
    close(oracle.dtrmm(side, uplo, trans, diag, 1.25, T, B.copy(order="F")), ref, nt)
    ref = sb.dtrsm(0.5, T, B, side=side, lower=1 - uplo, trans_a=trans, diag=diag)
    close(oracle.dtrsm(side, uplo, trans, diag, 0.5, T, B.copy(order="F")), ref, nt)


@pytest.mark.parametrize("uplo,trans", list(itertools.product((0, 1), (0, 1))))
def test_dsyrk_touches_one_triangle(oracle, uplo, trans):
    n, k = 210, 77
    A = rnd(k, n) if trans else rnd(n, k)
    C0 = rnd(n, n)
    ref = sb.dsyrk(-1.0, A, beta=1.0, c=C0, trans=trans, lower=1 - uplo)
    got = oracle.dsyrk(uplo, trans, -1.0, A, 1.0, C0.copy(order="F"))
    tri = np.triu(np.ones((n, n), bool)) if uplo else np.tril(np.ones((n, n), bool))
    close(got[tri], ref[tri], k)
    np.testing.assert_array_equal(got[~tri], C0[~tri])


@pytest.mark.parametrize("uplo", (0, 1))
@pytest.mark.parametrize("n", [1, 5, 96, 97, 400])
def test_dpotrf_dtrtri(oracle, uplo, n):
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    ref, info = sl.dpotrf(A, lower=1 - uplo, clean=0)
    got = A.copy(order="F")
    assert oracle.dpotrf(uplo, got) == info == 0
    tri = np.triu(np.ones((n, n), bool)) if uplo else np.tril(np.ones((n, n), bool))
    close(got[tri], ref[tri], n)
    np.testing.assert_array_equal(got[~tri], A[~tri])          # LAPACK leaves the other triangle alone
    inv_ref, info = sl.dtrtri(ref, lower=1 - uplo)
    inv = got.copy(order="F")
    assert oracle.dtrtri(uplo, 0, inv) == info == 0
    close(inv[tri], inv_ref[tri], n)


def test_dpotrf_info(oracle):
    A = oracle.distribute_symmetric(40, 40, 0, 0, 1, 1)
    A[25, 25] = -1.0
    _, info = sl.dpotrf(A, lower=0)
    assert oracle.dpotrf(1, A.copy(order="F")) == info == 26


MKL_PATHS = ("/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so")


@pytest.mark.skipif(not any(os.path.exists(p) for p in MKL_PATHS), reason="libmkl_rt.so (the reference's third-party BLAS) is not installed here")
def test_against_mkl_entry_points():
    """The exact symbols the reference binds (src/blas/interface.hpp:54,74,92; src/lapack/interface.hpp:39,54), called
    with the CBLAS/LAPACKE constants it passes.  Runs in a fresh interpreter with MKL's sequential threading layer:
    libmkl_rt picks its interface/threading layers at load time and misbehaves next to an already-loaded OpenMP runtime."""
    import subprocess
    import sys
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", MKL_INTERFACE_LAYER="LP64", OMP_NUM_THREADS="1")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_mkl_check.py")
    r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MKL-OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("m,n", [(40, 40), (120, 33), (17, 29)])
def test_oracle_householder_qr(oracle, m, n):
    """orc_dgeqrf / orc_dorgqr (dgeqr2 / dorg2r restated) against numpy's LAPACK-backed QR: |R| agrees (the sign convention
    of the reflectors is LAPACK's own, so R matches numpy's 'r' mode exactly up to rounding), Q^T Q = I and Q R = A."""
    rng = np.random.default_rng(m * 100 + n)
    A = np.asfortranarray(rng.random((m, n)) - 0.5)
    F = A.copy(order="F")
    tau = oracle.dgeqrf(F)
    k = min(m, n)
    R = np.triu(F[:k, :])
    Rn = np.linalg.qr(A, mode="r")
    assert np.abs(R - Rn).max() <= 1e-13
    if m >= n:
        Q = F.copy(order="F")
        oracle.dorgqr(Q, tau)
        assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-14
        assert np.abs(Q @ R - A).max() <= 1e-14
