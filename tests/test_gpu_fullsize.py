"""BASELINE configs 2 and 3 at FULL size (n = 32768 recursive Cholesky; CA-CholeskyQR2 m = 2^22, n = 256), the metric's own matrix
(n = 65536 on one GPU) and the per-GPU slice
of config 5 (CA-CholeskyQR2 on 2^23 x 1024: what each of the 8 GPUs holds of m = 2^26), checked through
size-independent properties -- the oracle cannot reach these sizes in test time:
  * the reference's own validators (test/cholesky/validate.hpp, test/qr/validate.hpp) at the tolerances of SURVEY.md 8c
    (Cholesky residual <= 1e-14, CQR2 residual <= 1e-14, orthogonality <= 1e-15);
  * structure: R and R^-1 upper triangular, positive diagonal, the skipped R^-1_12 block exactly zero (complete_inv = 0);
  * inverse round trip on both diagonal halves: R_ii * R^-1_ii = I;
  * for QR: unit column norms of Q, A = Q R on a random sample of rows, R upper triangular with positive diagonal.
All device math below is plain torch fp64 on data fetched through the driver's C-ABI (the product never sees torch)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def drv():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    from capital_amd import driver
    driver.init(0, 0, 1, None, use_torch_stream=False)
    yield driver
    driver.finalize()


@pytest.mark.gpu
def test_cholinv_config2_full_size_properties(drv):
    import torch
    n, h = 32768, 16384
    prob = drv.Cholinv(n, c=1, complete_inv=0, split=1, bc_mult=-5, serialize=True, bc_policy=2)
    prob.generate()
    prob.factor()
    drv.sync()
    assert prob.residual() <= 1e-14                                   # ||A - R^T R||_F / ||A||_F, reference validator
    R = torch.from_numpy(prob.R()).cuda()                             # column-major (n, n) numpy -> tensor R[i, j]
    Ri = torch.from_numpy(prob.Rinv()).cuda()
    prob.close()
    assert torch.count_nonzero(torch.tril(R, -1)).item() == 0 and torch.count_nonzero(torch.tril(Ri, -1)).item() == 0
    assert (torch.diagonal(R) > 0).all() and (torch.diagonal(Ri) > 0).all()
    assert torch.count_nonzero(Ri[:h, h:]).item() == 0                # top-level R^-1_12 is not completed
    eye = torch.eye(h, dtype=torch.float64, device="cuda")
    for s in (slice(0, h), slice(h, n)):
        err = (R[s, s] @ Ri[s, s] - eye).abs().max().item()
        assert err <= 1e-11, err                                      # kappa(R_ii) is small: the generator adds n to the diagonal
    # diagonal of R^T R reproduces diag(A) = n + U[0,1): cheap independent look at the factor itself
    d = (R * R).sum(dim=0)
    assert ((d >= n - 1e-6) & (d <= n + 1 + 1e-6)).all()
    del R, Ri, d, eye
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_cholinv_n65536_properties(drv):
    """The headline's own matrix (BASELINE `metric`: Cholesky n = 65536; on ONE MI355X: 139 GiB with both factors and their packed
    copies), through properties only: the reference's validator, structure, the diagonal of R^T R, and the inverse round trip
    R_ii R^-1_ii = I on both diagonal halves in 16384-blocks (diagonal blocks = I, the off-diagonal block R_aa X_ab + R_ab X_bb = 0)."""
    import torch
    n, h, b = 65536, 32768, 16384
    prob = drv.Cholinv(n, c=1, complete_inv=0, split=1, bc_mult=-6, serialize=True, bc_policy=2)
    prob.generate()
    prob.factor()
    drv.sync()
    st = prob.stats()
    assert st["bc_dimension"] == 1024
    res = prob.residual()
    assert res <= 1e-14, res                                          # ||A - R^T R||_F / ||A||_F (test/cholesky/validate.hpp:7-49)
    R = torch.from_numpy(prob.R()).cuda()
    Ri = torch.from_numpy(prob.Rinv()).cuda()
    prob.close()
    from capital_amd import capi
    capi.load().capi_trim_workspaces(drv.handle_ptr())                # the arena is gone with `prob`; this frees the kernels' own scratch
    assert (torch.diagonal(R) > 0).all() and (torch.diagonal(Ri) > 0).all()
    for k in range(0, n, b):                                          # strictly-lower parts are exactly zero (block by block: no n x n temporaries)
        assert torch.count_nonzero(torch.tril(R[k:k + b, k:k + b], -1)).item() == 0
        assert torch.count_nonzero(torch.tril(Ri[k:k + b, k:k + b], -1)).item() == 0
        if k:
            assert torch.count_nonzero(R[k:k + b, :k]).item() == 0 and torch.count_nonzero(Ri[k:k + b, :k]).item() == 0
    assert torch.count_nonzero(Ri[:h, h:]).item() == 0                # top-level R^-1_12 is not completed (complete_inv = 0)
    d = torch.zeros(n, dtype=torch.float64, device="cuda")
    for k in range(0, n, b):
        d[k:k + b] = (R[:k + b, k:k + b] * R[:k + b, k:k + b]).sum(dim=0)
    assert ((d >= n - 1e-6) & (d <= n + 1 + 1e-6)).all()              # diag(R^T R) = diag(A) = n + U[0,1)
    eye = torch.eye(b, dtype=torch.float64, device="cuda")
    for o in (0, h):
        a0, a1, b0, b1 = o, o + b, o + b, o + 2 * b
        assert (R[a0:a1, a0:a1] @ Ri[a0:a1, a0:a1] - eye).abs().max().item() <= 1e-11
        assert (R[b0:b1, b0:b1] @ Ri[b0:b1, b0:b1] - eye).abs().max().item() <= 1e-11
        off = R[a0:a1, a0:a1] @ Ri[a0:a1, b0:b1] + R[a0:a1, b0:b1] @ Ri[b0:b1, b0:b1]
        assert off.abs().max().item() <= 1e-11
        del off
    del R, Ri, d, eye
    torch.cuda.empty_cache()                                          # 64 GiB back to the device: the config-5 slice below needs 192 GiB


@pytest.mark.gpu
def test_cacqr2_config3_full_size_properties(drv):
    import torch
    m, n = 1 << 22, 256
    q = drv.Cacqr(m, n, c=1, variant=2)
    q.generate()
    q.factor()
    drv.sync()
    assert q.residual() <= 1e-14 and q.orthogonality() <= 1e-15       # reference validators, SURVEY 8c tolerances
    R = q.R()
    assert np.count_nonzero(np.tril(R, -1)) == 0 and (np.diag(R) > 0).all()
    Q = torch.from_numpy(q.Q()).cuda()
    A = q.A()
    q.close()
    assert (torch.linalg.vector_norm(Q, dim=0) - 1.0).abs().max().item() <= 1e-13
    G = Q.T @ Q
    assert (G - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max().item() <= 1e-13
    rows = np.random.default_rng(7).integers(0, m, size=4096)
    back = Q[torch.from_numpy(rows).cuda()].cpu().numpy() @ R
    assert np.abs(back - A[rows]).max() <= 1e-12 * np.abs(A[rows]).max() * n


@pytest.mark.gpu
def test_cacqr2_config5_slice(drv):
    """BASELINE config 5 on one of its 8 GPUs: the local 2^23 x 1024 row block (64 GiB; A + Q's two buffers = 192 GiB of the
    288 GB).  With one rank the Gram all-reduce is the identity, so this is the whole of what a GPU computes in config 5."""
    m, n = 1 << 23, 1024
    q = drv.Cacqr(m, n, c=1, variant=2)
    q.generate()
    q.factor()
    drv.sync()
    res, orth = q.residual(), q.orthogonality()
    assert res <= 1e-14 and orth <= 1e-15, (res, orth)               # reference validators, SURVEY 8c tolerances
    R = q.R()
    assert np.count_nonzero(np.tril(R, -1)) == 0 and (np.diag(R) > 0).all()
    G = q.gram_of_Q()                                                 # Q^T Q: unit column norms on its diagonal
    assert np.abs(np.diag(G) - 1.0).max() <= 1e-13
    assert np.abs(G - np.eye(n)).max() <= 1e-13
    # A = Q R on row windows spread over the panel (first, last, and random interior ones)
    rng = np.random.default_rng(11)
    starts = [0, m - 512] + [int(s) for s in rng.integers(0, m - 512, size=6)]
    for s0 in starts:
        Aw, Qw = q.rows("A", s0, 512), q.rows("Q", s0, 512)
        assert np.abs(Qw @ R - Aw).max() <= 1e-12 * n, s0
    # the generator's stream position: element (i, j) of the single-rank panel is draw number j m + i + 1 of srand48(0)
    assert 0.0 <= q.rows("A", m - 1, 1).min() and q.rows("A", m - 1, 1).max() < 1.0
    q.close()
