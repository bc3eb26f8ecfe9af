"""CPU: the oracle's cholinv / 1-D cacqr schedules through the reference's own validator metrics
(test/cholesky/validate.hpp, test/qr/validate.hpp) and the values SURVEY.md section 4 recorded for the unmodified
reference: Cholesky residual 0.9-1.9e-16, CQR2 residual 5.9e-16, orthogonality 1.6-2.1e-17."""
import numpy as np
import pytest
import scipy.linalg as sl


@pytest.mark.parametrize("n", (512, 1000, 1024))
@pytest.mark.parametrize("bc", (0, -1, -2, -3))
@pytest.mark.parametrize("ci", (0, 1))
def test_cholinv_residual_band(oracle, n, bc, ci):
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    R, Ri, info = oracle.cholinv_factor(A, ci, 1, bc, 1, 1)
    assert info == 0
    res = oracle.cholesky_residual(A, R)
    assert res <= 6e-16, res                       # same order as the reference's 0.9-1.9e-16
    assert np.abs(R - sl.cholesky(A, lower=False)).max() <= 1e-12 * np.abs(R).max()
    assert np.all(np.tril(R, -1) == 0) and np.all(np.tril(Ri, -1) == 0)
    if ci or bc == 0:
        assert np.abs(Ri @ R - np.eye(n)).max() <= 1e-13
    else:
        h = n // 2                                 # complete_inv = 0: block-diagonal inverse at the top level only
        assert np.all(Ri[:h, h:] == 0)
        assert np.abs(Ri[:h, :h] @ R[:h, :h] - np.eye(h)).max() <= 1e-13
        assert np.abs(Ri[h:, h:] @ R[h:, h:] - np.eye(n - h)).max() <= 1e-13


@pytest.mark.parametrize("c,d", [(1, 1), (2, 2), (1, 2), (2, 1), (3, 3)])
def test_cholinv_is_grid_invariant(oracle, c, d):
    """R is unique: the collapsed-grid recursion must land on the same factor for every grid and base-case size,
    including grids that do not divide n (padding path)."""
    n = 250
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    ref = sl.cholesky(A, lower=False)
    for bc in (1, 0, -1, -2):
        R, Ri, info = oracle.cholinv_factor(A, 1, 1, bc, c, d)
        assert info == 0 and np.abs(R - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.abs(Ri @ R - np.eye(n)).max() <= 1e-12


def test_base_case_rule(oracle):
    """cholinv.hpp:15-18 with the sizes SURVEY.md 8a row A3 quotes."""
    assert oracle.cholinv_bc_dimension(32768, 1, 1, -4) == 2048        # config (2), bc = -4
    assert oracle.cholinv_bc_dimension(32768, 2, 2, 0) == 16384        # config (4), bc = 0: t = c*d = 4
    assert oracle.cholinv_bc_dimension(32768, 2, 2, -3) == 2048        # config (4), bc = -3
    assert oracle.cholinv_bc_dimension(1024, 1, 1, 0) == 1024          # 1 rank, bc >= 0: no recursion
    assert oracle.cholinv_bc_dimension(1024, 1, 1, 3) == 1024
    assert oracle.cholinv_bc_dimension(100, 1, 1, -20) == 1            # clamp to [1, n_loc]


@pytest.mark.parametrize("m,n", [(16384, 256), (4097, 33)])
@pytest.mark.parametrize("P", (1, 8))
def test_cacqr2_bands(oracle, m, n, P):
    A = oracle.distribute_random(n, m, 0, 0, 1, 1, key=0)
    Q, R, info = oracle.cacqr_factor_1d(A, P, 2)
    assert info == 0
    assert oracle.qr_residual(A, Q, R) <= 2e-15         # reference: 5.9e-16
    assert oracle.qr_orthogonality(Q) <= 2e-16          # reference: 1.6-2.1e-17
    assert np.all(np.tril(R, -1) == 0) and np.all(np.diag(R) > 0)
    Q1, R1, _ = oracle.cacqr_factor_1d(A, P, 1)         # one sweep: orthogonality ~1e-15 (SURVEY section 4 variant 1)
    assert oracle.qr_orthogonality(Q1) <= 1e-13 and oracle.qr_residual(A, Q1, R1) <= 1e-14


def test_serialize_layouts(oracle):
    """packed offsets (structure.h:39,59) and the copy counts of serialize.hpp on a small case, by hand"""
    import ctypes as C
    n = 5
    assert [oracle.lib().orc_offset(1, x, 0, n, n) for x in range(n)] == [0, 1, 3, 6, 10]
    assert [oracle.lib().orc_offset(2, x, x, n, n) for x in range(n)] == [0, 5, 9, 12, 14]
    full = np.asfortranarray(np.arange(25, dtype=float).reshape(5, 5).T)   # full[y, x] = x*5 + y
    packed = np.zeros(15)
    dp = C.POINTER(C.c_double)
    oracle.lib().orc_serialize(0, 1, full.ctypes.data_as(dp), n, n, packed.ctypes.data_as(dp), n, n, 0, n, 0, n, 0, n, 0, n)
    assert list(packed) == [float(x * 5 + y) for x in range(5) for y in range(x + 1)]
