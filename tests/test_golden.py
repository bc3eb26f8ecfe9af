"""Golden fixtures (tests/golden/capital_golden.npz, made by tests/golden/make_golden.py from glibc drand48 and
LAPACK) against the oracle on CPU and against the HIP path on MI355X.  R of an SPD matrix and the Q,R of a full-rank
matrix with positive diagonal are unique, so every schedule parameter must land on the same fixture."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "capital_golden.npz"))


@pytest.mark.parametrize("n", (64, 96))
def test_oracle_generator_and_cholinv_match_golden(oracle, n):
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    np.testing.assert_array_equal(A, G[f"spd_{n}"])                       # generator: bit exact
    for bc, ci, split in ((0, 0, 1), (-1, 1, 1), (-2, 0, 1), (-3, 1, 1), (-2, 1, 2)):
        R, Ri, info = oracle.cholinv_factor(A, ci, split, bc, 1, 1)
        assert info == 0
        assert np.abs(R - G[f"R_{n}"]).max() <= 1e-12 * np.abs(G[f"R_{n}"]).max()
        ref = G[f"Rinv_{n}"].copy()
        if not ci and oracle.cholinv_bc_dimension(n, 1, 1, bc) < n:      # top-level off-diagonal block is skipped (cholinv.hpp:147)
            h = n >> split
            ref[:h, h:] = 0.0
        assert np.abs(Ri - ref).max() <= 1e-12 * np.abs(ref).max()


def test_oracle_cacqr_matches_golden(oracle):
    A = oracle.distribute_random(24, 512, 0, 0, 1, 1, key=0)
    np.testing.assert_array_equal(A, G["tall_512x24"])
    for P in (1, 2, 5):
        Q, R, info = oracle.cacqr_factor_1d(A, P, 2)
        assert info == 0
        assert np.abs(R - G["Rq_512x24"]).max() <= 1e-12 * np.abs(G["Rq_512x24"]).max()
        assert np.abs(Q - G["Q_512x24"]).max() <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("n", (64, 96))
def test_gpu_cholinv_matches_golden(n):
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    try:
        for bc, ci, ser in ((0, 0, True), (-1, 1, False), (-1, 0, True)):
            p = driver.Cholinv(n, complete_inv=ci, bc_mult=bc, serialize=ser)
            p.generate()
            np.testing.assert_array_equal(p.A(), G[f"spd_{n}"])
            p.factor()
            assert np.abs(p.R() - G[f"R_{n}"]).max() <= 1e-12 * np.abs(G[f"R_{n}"]).max()
            ref = G[f"Rinv_{n}"].copy()
            if not ci and bc < 0:
                ref[: n // 2, n // 2:] = 0.0
            assert np.abs(p.Rinv() - ref).max() <= 1e-12 * np.abs(ref).max()
            p.close()
    finally:
        driver.finalize()


@pytest.mark.gpu
def test_gpu_cacqr_matches_golden():
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    try:
        q = driver.Cacqr(512, 24, variant=2)
        q.generate()
        np.testing.assert_array_equal(q.A(), G["tall_512x24"])
        q.factor()
        assert np.abs(q.R() - G["Rq_512x24"]).max() <= 1e-12 * np.abs(G["Rq_512x24"]).max()
        assert np.abs(q.Q() - G["Q_512x24"]).max() <= 1e-12
        q.close()
    finally:
        driver.finalize()


# ---- whole schedules on the reference's own arithmetic provider (MKL's cblas_*/LAPACKE_* entry points) ------------------------
M = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "capital_mkl_schedules.npz"))
CHOL_CASES = [(160, 0, 1, -2), (192, 1, 1, -3), (130, 1, 2, -2), (96, 1, 1, 0)]
CQR_CASES = [(1536, 48, 2), (1000, 24, 1)]


def _tri(n, packed):
    T = np.zeros((n, n))
    T[np.triu_indices(n)] = packed
    return T


@pytest.mark.parametrize("n,ci,split,bc", CHOL_CASES)
def test_oracle_cholinv_matches_mkl_schedule(oracle, n, ci, split, bc):
    """the oracle's own kernels against the same schedule run on MKL (tests/golden/make_mkl_golden.py)"""
    key = f"chol_n{n}_ci{ci}_s{split}_bc{-bc}"
    R, Ri, info = oracle.cholinv_factor(oracle.distribute_symmetric(n, n, 0, 0, 1, 1), ci, split, bc, 1, 1)
    assert info == 0 and not oracle.host_blas_active()
    for got, ref in ((R, _tri(n, M[key + "_R"])), (Ri, _tri(n, M[key + "_Rinv"]))):
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.count_nonzero(got) == np.count_nonzero(ref)


@pytest.mark.parametrize("m,n,variant", CQR_CASES)
def test_oracle_cacqr_matches_mkl_schedule(oracle, m, n, variant):
    key = f"cqr_m{m}_n{n}_v{variant}"
    Q, R, info = oracle.cacqr_factor_1d(oracle.distribute_random(n, m, 0, 0, 1, 1, key=0), 1, variant)
    assert info == 0
    Rref = _tri(n, M[key + "_R"])
    assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max()
    assert np.abs(Q - M[key + "_Q"]).max() <= 1e-12 * (1 if variant == 2 else 50)


@pytest.mark.gpu
@pytest.mark.parametrize("n,ci,split,bc", CHOL_CASES)
def test_gpu_cholinv_matches_mkl_schedule(n, ci, split, bc):
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    try:
        key = f"chol_n{n}_ci{ci}_s{split}_bc{-bc}"
        for serialize in (True, False):
            p = driver.Cholinv(n, c=1, complete_inv=ci, split=split, bc_mult=bc, serialize=serialize, bc_policy=2)
            p.generate()
            p.factor()
            for got, ref in ((p.R(), _tri(n, M[key + "_R"])), (p.Rinv(), _tri(n, M[key + "_Rinv"]))):
                assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
                assert np.count_nonzero(got) == np.count_nonzero(ref)
            p.close()
    finally:
        driver.finalize()


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,variant", CQR_CASES)
def test_gpu_cacqr_matches_mkl_schedule(m, n, variant):
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    try:
        key = f"cqr_m{m}_n{n}_v{variant}"
        q = driver.Cacqr(m, n, c=1, variant=variant)
        q.generate()
        q.factor()
        Rref = _tri(n, M[key + "_R"])
        assert np.abs(q.R() - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(q.Q() - M[key + "_Q"]).max() <= 1e-12 * (1 if variant == 2 else 50)
        q.close()
    finally:
        driver.finalize()
