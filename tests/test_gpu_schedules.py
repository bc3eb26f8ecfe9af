"""GPU parity of the two schedules through the C-ABI driver (capital_amd/drivers): recursive Cholesky-with-inverse
(cholinv.hpp:6-183) and 1-D CholeskyQR2 (cacqr.hpp:7-29,174-193) against the CPU oracle on the reference's own
generators.  Tolerances are the ones SURVEY.md 8c states: elementwise 1e-12 relative for R / R^-1 / Q, Cholesky
residual <= 1e-14, CQR2 residual <= 1e-14, orthogonality <= 1e-15."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drv():
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    yield driver
    driver.finalize()


@pytest.mark.parametrize("serialize", (True, False))
@pytest.mark.parametrize("n,bc,ci,split", [(256, 0, 0, 1), (512, -1, 0, 1), (512, -2, 1, 1), (1000, -3, 0, 1), (1024, -4, 1, 1), (768, -3, 1, 2), (2048, -5, 0, 1)])
def test_cholinv_matches_oracle(drv, oracle, n, bc, ci, split, serialize):
    p = drv.Cholinv(n, c=1, complete_inv=ci, split=split, bc_mult=bc, serialize=serialize, bc_policy=2)
    p.generate()
    A = p.A()
    np.testing.assert_array_equal(A, oracle.distribute_symmetric(n, n, 0, 0, 1, 1))     # same input, bit for bit
    p.factor()
    R, Ri = p.R(), p.Rinv()
    Rref, Riref, info = oracle.cholinv_factor(A, ci, split, bc, 1, 1)
    assert info == 0
    assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max()
    assert np.abs(Ri - Riref).max() <= 1e-12 * np.abs(Riref).max()
    # structure: strictly lower parts are zero; without complete_inv the top-level off-diagonal block of R^-1 stays zero
    assert np.all(np.tril(R, -1) == 0) and np.all(np.tril(Ri, -1) == 0)
    assert (np.count_nonzero(Ri) == np.count_nonzero(Riref))
    res = p.residual()
    assert res <= 1e-14 and abs(res - oracle.cholesky_residual(A, Rref)) <= 5e-16
    st = p.stats()
    assert st["bc_dimension"] == oracle.cholinv_bc_dimension(n, 1, 1, bc)
    p.close()


@pytest.mark.parametrize("n,bc,ci,split", [(1000, -3, 0, 1), (1024, -4, 1, 1), (768, -3, 1, 2), (2048, -5, 0, 1), (4608, -3, 1, 1)])
def test_cholinv_lookahead_matches_oracle(drv, oracle, monkeypatch, n, bc, ci, split):
    """single-GPU lookahead (the bulk of each trailing update on a second stream beside the next block's factorisation),
    forced on at every level that splits; the default threshold (2048) only reaches it at n >= 4096"""
    monkeypatch.setenv("CAPITAL_LOOKAHEAD_MIN", "64")
    p = drv.Cholinv(n, c=1, complete_inv=ci, split=split, bc_mult=bc, serialize=True, bc_policy=2)
    p.generate()
    A = p.A()
    p.factor()
    R, Ri = p.R(), p.Rinv()
    p.factor()                                           # second call: streams, events and workspaces are reused
    np.testing.assert_array_equal(R, p.R())
    monkeypatch.setenv("CAPITAL_NO_LOOKAHEAD", "1")
    p.factor()
    R0, Ri0 = p.R(), p.Rinv()
    Rref, Riref, info = oracle.cholinv_factor(A, ci, split, bc, 1, 1)
    assert info == 0
    for got, got0, ref in ((R, R0, Rref), (Ri, Ri0, Riref)):
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
        assert np.abs(got - got0).max() <= 1e-13 * np.abs(ref).max()
        assert np.all(np.tril(got, -1) == 0) and np.count_nonzero(got) == np.count_nonzero(ref)
    assert p.residual() <= 1e-14
    p.close()


def test_cholinv_lookahead_random_orders(drv, oracle, monkeypatch):
    """random orders / base-case depths / split shifts with the lookahead forced on at every level (8 cases by default;
    CAPITAL_FUZZ_CASES widens the sweep for one-off runs)"""
    import os
    rng = np.random.default_rng(int(os.environ.get("CAPITAL_FUZZ_SEED", "7")))
    monkeypatch.setenv("CAPITAL_LOOKAHEAD_MIN", "32")
    for _ in range(int(os.environ.get("CAPITAL_FUZZ_CASES", "8"))):
        n = int(rng.integers(200, 1800))
        bc, ci, split = -int(rng.integers(1, 5)), int(rng.integers(2)), int(rng.choice([1, 1, 1, 2]))
        p = drv.Cholinv(n, c=1, complete_inv=ci, split=split, bc_mult=bc, serialize=bool(rng.integers(2)), bc_policy=2)
        p.generate()
        A = p.A()
        p.factor()
        R, Ri = p.R(), p.Rinv()
        Rref, Riref, info = oracle.cholinv_factor(A, ci, split, bc, 1, 1)
        assert info == 0
        assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max(), (n, bc, ci, split)
        assert np.abs(Ri - Riref).max() <= 1e-12 * np.abs(Riref).max(), (n, bc, ci, split)
        assert np.count_nonzero(Ri) == np.count_nonzero(Riref)
        p.close()


def test_cholinv_repeat_is_deterministic(drv, oracle):
    p = drv.Cholinv(768, bc_mult=-2, serialize=False)
    p.generate()
    p.factor()
    R1 = p.R()
    p.factor()
    np.testing.assert_array_equal(R1, p.R())
    p.close()


@pytest.mark.parametrize("serialize", (True, False))
@pytest.mark.parametrize("m,n,variant", [(4096, 64, 2), (65536, 256, 2), (50001, 130, 2), (8192, 32, 1), (20000, 512, 2)])
def test_cacqr_1d_matches_oracle(drv, oracle, m, n, variant, serialize):
    p = drv.Cacqr(m, n, c=1, variant=variant, serialize=serialize)
    p.generate()
    A = p.A()
    np.testing.assert_array_equal(A, oracle.distribute_random(n, m, 0, 0, 1, 1, key=0))
    p.factor()
    Q, R = p.Q(), p.R()
    Qref, Rref, info = oracle.cacqr_factor_1d(A, 1, variant)
    assert info == 0
    assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max()
    assert np.abs(Q - Qref).max() <= 1e-12 * max(1.0, np.abs(Qref).max()) * (1 if variant == 2 else 50)
    assert np.all(np.tril(R, -1) == 0)
    if variant == 2:
        assert p.residual() <= 1e-14
        assert p.orthogonality() <= 1e-15
        assert abs(p.residual() - oracle.qr_residual(A, Qref, Rref)) <= 5e-16
    p.close()


def test_cacqr2_width_1024_matches_oracle(drv, oracle):
    """BASELINE config 5's width (n = 1024) at a height the oracle finishes in seconds: Q, R elementwise against
    oracle.cacqr_factor_1d (cacqr.hpp:7-29,174-193) on the reference's generator, plus the reference validators."""
    m, n = 1 << 17, 1024
    p = drv.Cacqr(m, n, c=1, variant=2, serialize=True)
    p.generate()
    A = p.A()
    np.testing.assert_array_equal(A, oracle.distribute_random(n, m, 0, 0, 1, 1, key=0))
    p.factor()
    Q, R = p.Q(), p.R()
    Qref, Rref, info = oracle.cacqr_factor_1d(A, 1, 2)
    assert info == 0
    assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max()
    assert np.abs(Q - Qref).max() <= 1e-12 * max(1.0, np.abs(Qref).max())
    assert np.all(np.tril(R, -1) == 0) and (np.diag(R) > 0).all()
    assert p.residual() <= 1e-14 and p.orthogonality() <= 1e-15
    assert abs(p.residual() - oracle.qr_residual(A, Qref, Rref)) <= 5e-16
    # the row-window accessor and Q^T Q (used by the full-size test) agree with the whole-panel getters
    np.testing.assert_array_equal(p.rows("Q", 1000, 77), Q[1000:1077])
    np.testing.assert_array_equal(p.rows("A", m - 5, 5), A[m - 5:])
    assert np.abs(p.gram_of_Q() - Q.T @ Q).max() <= 1e-13
    p.close()


def test_factor_reports_a_non_spd_input(drv, oracle):
    """The reference drops LAPACK's info (lapack/interface.hpp:39,54) and returns garbage; here the device-side info word is
    read once at the end of factor() and a failed pivot raises -- for the recursive Cholesky and for CholeskyQR's Gram matrix."""
    from capital_amd.driver import DriverError
    n = 512
    p = drv.Cholinv(n, bc_mult=-2, serialize=False)
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    A[300, 300] = -1.0                                    # indefinite: the pivot at global index 300 fails
    p.set_A(A)
    with pytest.raises(DriverError, match="not positive definite"):
        p.factor()
    p.set_A(oracle.distribute_symmetric(n, n, 0, 0, 1, 1))
    p.factor()                                            # the handle recovers: the next call starts from a clean info word
    assert p.residual() <= 1e-14
    p.close()
    q = drv.Cacqr(4096, 64, c=1, variant=2)
    Aq = oracle.distribute_random(64, 4096, 0, 0, 1, 1, key=0)
    Aq[:, 10] = 0.0                                       # a zero column: A^T A has an exactly zero pivot
    q.set_A(Aq)
    with pytest.raises(DriverError, match="not positive definite|rank deficient"):
        q.factor()
    q.close()


@pytest.mark.parametrize("serialize", (True, False))
@pytest.mark.parametrize("n,bc,split", [(512, -1, 1), (1000, -3, 1), (2048, -3, 1), (768, -3, 2), (4608, -3, 1)])
def test_cholinv_trsm_mode_gives_the_same_R(drv, oracle, n, bc, split, serialize):
    """TRSM mode (info::solve_with_trsm; BASELINE north_star's POTRF + block TRSM + SYRK): the same factor R as the
    reference-exact schedule and the oracle, no inverse formed."""
    from capital_amd.driver import DriverError
    p = drv.Cholinv(n, c=1, complete_inv=0, split=split, bc_mult=bc, serialize=serialize, trsm_mode=True)
    p.generate()
    A = p.A()
    p.factor()
    R = p.R()
    Rref, _, info = oracle.cholinv_factor(A, 0, split, bc, 1, 1)
    assert info == 0
    assert np.abs(R - Rref).max() <= 1e-12 * np.abs(Rref).max()
    assert np.all(np.tril(R, -1) == 0)
    assert p.residual() <= 1e-14
    assert p.stats()["bc_dimension"] == oracle.cholinv_bc_dimension(n, 1, 1, bc)
    with pytest.raises(DriverError, match="no inverse"):
        p.Rinv()
    p.close()


def test_flush_intermediates_releases_the_working_images(drv, oracle):
    """FlushIntermediates (policy.h:85-156) as a memory policy: after factor() only the results stay on the device -- with
    Serialize that is the two packed triangles, not the full-storage working images and the arena -- and the factors are the same.
    (Device memory is read after the validator has run: the runtime reports freed blocks with a delay.)"""
    import torch
    n, bc = 4096, -2
    blk = 8 * n * n

    def footprint(flush):
        drv.sync()
        free0 = torch.cuda.mem_get_info()[0]
        p = drv.Cholinv(n, bc_mult=bc, serialize=True, flush_intermediates=flush)
        p.generate()
        p.factor()
        p.factor()                                        # a second call re-creates what the first released
        assert p.residual() <= 1e-14
        R = p.R()
        drv.sync()
        used = free0 - torch.cuda.mem_get_info()[0]
        p.close()
        drv.sync()
        return used, R, free0 - torch.cuda.mem_get_info()[0]

    footprint(False)                                      # (the handle's own workspaces reach their final size)
    used_keep, Rk, left_keep = footprint(False)
    used_flush, Rf, left_flush = footprint(True)
    np.testing.assert_array_equal(Rf, Rk)
    # Save: A + 2 packed triangles + 2 full images + arena (4.33 blocks);  Flush: A + 2 packed triangles (2 blocks)
    assert used_keep - used_flush >= 2 * blk, (used_keep, used_flush)
    assert abs(left_keep) <= blk // 16 and abs(left_flush) <= blk // 16, (left_keep, left_flush)      # nothing leaks past close()
