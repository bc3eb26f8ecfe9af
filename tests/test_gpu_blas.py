"""GPU parity: BLAS-level C-ABI (capi_dgemm/dgemmt/dsyrk/dtrmm/dtrsm) against the CPU oracle's
restatement of cblas_dgemm/dsyrk/dtrmm (reference src/blas/interface.hpp:43-97).
Tolerance: fp64, |gpu - oracle| <= 1e-13 * k * max|entries| (summation-order differences only)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(rng, m, n):
    return np.asfortranarray(rng.uniform(-1, 1, size=(m, n)))


def _check(got, ref, k, scale=1.0):
    tol = 1e-14 * max(k, 16) * max(scale, 1.0)
    err = np.abs(got - ref).max() if got.size else 0.0
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


def test_mfma_identity_layout(hip, oracle):
    """A = I with an ASYMMETRIC B: catches a transposed accumulator map (f64 MFMA C/D layout)."""
    from capital_amd import capi
    n = 128
    A = np.asfortranarray(np.eye(n))
    B = np.asfortranarray(np.arange(n * n, dtype=np.float64).reshape(n, n) % 97 + np.arange(n)[:, None] * 0.5)
    dA, dB, dC = capi.to_device(A), capi.to_device(B), capi.zeros(n, n)
    hip.call("capi_dgemm", 0, 0, n, n, n, 1.0, capi.ptr(dA), n, capi.ptr(dB), n, 0.0, capi.ptr(dC), n)
    np.testing.assert_array_equal(capi.to_host(dC), B)
    hip.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(dB), n, capi.ptr(dA), n, 0.0, capi.ptr(dC), n)
    np.testing.assert_array_equal(capi.to_host(dC), B.T)


@pytest.mark.parametrize("ta,tb", list(itertools.product((0, 1), (0, 1))))
@pytest.mark.parametrize("m,n,k,pad", [(128, 128, 64, 0), (200, 150, 77, 0), (257, 131, 1000, 3), (1, 1, 1, 0), (64, 300, 16, 1), (513, 129, 130, 0)])
def test_dgemm(hip, oracle, ta, tb, m, n, k, pad):
    from capital_amd import capi
    rng = np.random.default_rng(m * 7 + n * 3 + k + ta * 2 + tb)
    ar, ac = (k, m) if ta else (m, k)
    br, bc = (n, k) if tb else (k, n)
    A = _rand(rng, ar + pad, ac)[:ar, :]          # ld = ar + pad
    B = _rand(rng, br + pad, bc)[:br, :]
    Cm = _rand(rng, m + pad, n)[:m, :]
    Af, Bf, Cf = _rand(rng, ar + pad, ac), _rand(rng, br + pad, bc), _rand(rng, m + pad, n)
    A, B, Cm = Af[:ar, :], Bf[:br, :], Cf[:m, :]
    ref = np.asfortranarray(Cm.copy())
    oracle.dgemm(ta, tb, 0.75, np.asfortranarray(A), np.asfortranarray(B), -0.5, ref)
    dA, dB, dC = capi.to_device(Af), capi.to_device(Bf), capi.to_device(Cf)
    hip.call("capi_dgemm", ta, tb, m, n, k, 0.75, capi.ptr(dA), ar + pad, capi.ptr(dB), br + pad, -0.5, capi.ptr(dC), m + pad)
    got = capi.to_host(dC)
    _check(got[:m, :], ref, k)
    if pad:  # rows beyond m must be untouched
        np.testing.assert_array_equal(got[m:, :], Cf[m:, :])


def test_dgemm_beta0_ignores_nan(hip, oracle):
    from capital_amd import capi
    rng = np.random.default_rng(5)
    m = n = k = 96
    A, B = _rand(rng, m, k), _rand(rng, k, n)
    dA, dB, dC = capi.to_device(A), capi.to_device(B), capi.to_device(np.full((m, n), np.nan))
    hip.call("capi_dgemm", 0, 0, m, n, k, 1.0, capi.ptr(dA), m, capi.ptr(dB), k, 0.0, capi.ptr(dC), m)
    _check(capi.to_host(dC), A @ B, k)


def test_dgemm_unaligned_pointer(hip, oracle):
    """Odd element offsets defeat the 16-byte fast path; results must not change."""
    from capital_amd import capi
    rng = np.random.default_rng(6)
    m, n, k, ld = 130, 140, 150, 201
    Af, Bf = _rand(rng, ld, 256), _rand(rng, ld, 256)
    dA, dB, dC = capi.to_device(Af), capi.to_device(Bf), capi.zeros(m, n)
    offA, offB = 1 + 3 * ld, 5 + 1 * ld
    A = Af.ravel(order="F")[offA:offA + ld * k].reshape((ld, k), order="F")[:m, :]   # NoTrans m x k
    B = Bf.ravel(order="F")[offB:offB + ld * n].reshape((ld, n), order="F")[:k, :]
    hip.call("capi_dgemm", 0, 0, m, n, k, 1.0, capi.ptr(dA) + 8 * offA, ld, capi.ptr(dB) + 8 * offB, ld, 0.0, capi.ptr(dC), m)
    _check(capi.to_host(dC), A @ B, k)


@pytest.mark.parametrize("uplo", (0, 1))
@pytest.mark.parametrize("trans", (0, 1))
@pytest.mark.parametrize("n,k", [(128, 128), (300, 77), (1, 5), (515, 260)])
def test_dsyrk(hip, oracle, uplo, trans, n, k):
    from capital_amd import capi
    rng = np.random.default_rng(n + k + uplo + 2 * trans)
    A = _rand(rng, k, n) if trans else _rand(rng, n, k)
    Cm = _rand(rng, n, n)
    ref = Cm.copy(order="F")
    oracle.dsyrk(uplo, trans, -1.0, A, 1.0, ref)
    dA, dC = capi.to_device(A), capi.to_device(Cm)
    hip.call("capi_dsyrk", uplo, trans, n, k, -1.0, capi.ptr(dA), A.shape[0], 1.0, capi.ptr(dC), n)
    got = capi.to_host(dC)
    _check(got, ref, k)
    # the other triangle is untouched, bit for bit
    other = np.tril(np.ones((n, n), bool), -1) if uplo else np.triu(np.ones((n, n), bool), 1)
    np.testing.assert_array_equal(got[other], Cm[other])


def test_dsyrk_tall_skinny_splitk(hip, oracle):
    """CholeskyQR2's Gram matrix (cacqr.hpp:14-15): k = m_loc >> n, served by the split-K path; reproducible."""
    from capital_amd import capi
    rng = np.random.default_rng(11)
    k, n = 70001, 256
    A = _rand(rng, k, n)
    ref = np.zeros((n, n), order="F")
    oracle.dsyrk(1, 1, 1.0, A, 0.0, ref)
    dA = capi.to_device(A)
    outs = []
    for _ in range(2):
        dC = capi.to_device(np.full((n, n), np.nan))
        hip.call("capi_dsyrk", 1, 1, n, k, 1.0, capi.ptr(dA), k, 0.0, capi.ptr(dC), n)
        outs.append(capi.to_host(dC))
    iu = np.triu_indices(n)
    assert np.abs(outs[0][iu] - ref[iu]).max() <= 1e-14 * k
    np.testing.assert_array_equal(outs[0][iu], outs[1][iu])


@pytest.mark.parametrize("uplo,ta,tb", list(itertools.product((0, 1), (0, 1), (0, 1))))
def test_dgemmt(hip, oracle, uplo, ta, tb):
    from capital_amd import capi
    rng = np.random.default_rng(uplo * 4 + ta * 2 + tb)
    n, k = 333, 190
    A = _rand(rng, k, n) if ta else _rand(rng, n, k)
    B = _rand(rng, n, k) if tb else _rand(rng, k, n)
    Cm = _rand(rng, n, n)
    ref = Cm.copy(order="F")
    oracle.dgemm(ta, tb, 2.0, A, B, 0.5, ref)
    dA, dB, dC = capi.to_device(A), capi.to_device(B), capi.to_device(Cm)
    hip.call("capi_dgemmt", uplo, ta, tb, n, k, 2.0, capi.ptr(dA), A.shape[0], capi.ptr(dB), B.shape[0], 0.5, capi.ptr(dC), n)
    got = capi.to_host(dC)
    tri = np.triu(np.ones((n, n), bool)) if uplo else np.tril(np.ones((n, n), bool))
    _check(got[tri], ref[tri], k)
    np.testing.assert_array_equal(got[~tri], Cm[~tri])


@pytest.mark.parametrize("side,uplo,trans,diag", list(itertools.product((0, 1), (0, 1), (0, 1), (0, 1))))
@pytest.mark.parametrize("m,n", [(200, 130), (129, 257), (64, 64)])
def test_dtrmm(hip, oracle, side, uplo, trans, diag, m, n):
    from capital_amd import capi
    rng = np.random.default_rng(side * 8 + uplo * 4 + trans * 2 + diag + m)
    nt = m if side == 0 else n
    T = _rand(rng, nt, nt)  # full of junk in the unreferenced triangle on purpose
    B = _rand(rng, m, n)
    ref = B.copy(order="F")
    oracle.dtrmm(side, uplo, trans, diag, -1.5, T, ref)
    dT, dB, dCm = capi.to_device(T), capi.to_device(B), capi.zeros(m, n)
    hip.call("capi_dtrmm_oop", side, uplo, trans, diag, m, n, -1.5, capi.ptr(dT), nt, capi.ptr(dB), m, capi.ptr(dCm), m)
    _check(capi.to_host(dCm), ref, nt)
    hip.call("capi_dtrmm", side, uplo, trans, diag, m, n, -1.5, capi.ptr(dT), nt, capi.ptr(dB), m)
    _check(capi.to_host(dB), ref, nt)


@pytest.mark.parametrize("side,uplo,trans,diag", list(itertools.product((0, 1), (0, 1), (0, 1), (0, 1))))
@pytest.mark.parametrize("m,n", [(300, 70), (70, 300), (600, 520)])
def test_dtrsm(hip, oracle, side, uplo, trans, diag, m, n):
    from capital_amd import capi
    rng = np.random.default_rng(side * 8 + uplo * 4 + trans * 2 + diag + m)
    nt = m if side == 0 else n
    T = np.asfortranarray(_rand(rng, nt, nt) * 0.1 + np.eye(nt) * 4.0)   # well conditioned
    B = _rand(rng, m, n)
    ref = B.copy(order="F")
    oracle.dtrsm(side, uplo, trans, diag, 0.5, T, ref)
    dT, dB = capi.to_device(T), capi.to_device(B)
    hip.call("capi_dtrsm", side, uplo, trans, diag, m, n, 0.5, capi.ptr(dT), nt, capi.ptr(dB), m)
    got = capi.to_host(dB)
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,pad", [(20037, 256, 3), (65536, 256, 0), (30001, 192, 2), (16384 + 16, 256, 0)])
def test_tall_skinny_gram_and_right_trmm(m, n, pad):
    """The full-width tall-skinny kernels (Gram = capi_dsyrk with K = m >> n; Q = A T = capi_dtrmm_oop Right/Upper,
    cacqr.hpp:14-15,24-25) against a plain fp64 torch reference; ragged m, odd leading dimension, n below 256 (which
    falls back to the tile kernel for the TRMM).  Tolerance 1e-13 relative to the largest entry (sums of m products)."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    torch.manual_seed(m + n)
    ld = m + pad
    A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5           # column-major m x n, leading dimension ld
    Q = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5)
    Tcm = T.T.contiguous()                                                       # column-major storage of upper T
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), ld, 0.0, capi.ptr(G), n)
    h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(Tcm), n, capi.ptr(A), ld, capi.ptr(Q), ld)
    h.sync()
    Am = A[:, :m].T
    Gref = torch.triu(Am.T @ Am)
    Gout = torch.triu(G.T)
    Qref = Am @ T
    Qout = Q[:, :m].T
    assert (Gout - Gref).abs().max().item() <= 1e-13 * Gref.abs().max().item()
    assert (Qout - Qref).abs().max().item() <= 1e-13 * Qref.abs().max().item()
    assert torch.count_nonzero(Q[:, m:]).item() == 0                            # nothing written past row m
    assert torch.count_nonzero(torch.tril(G.T, -1)).item() == 0                  # lower triangle of G untouched


@pytest.mark.gpu
@pytest.mark.parametrize("m", [16384, 65536 + 32, 32 * 3001])
def test_panel32_images_match_column_major(m):
    """The "panel32" forms of the two tall-skinny kernels (include/capital_hip.h; qr::cacqr keeps CholeskyQR2's intermediate Q1 in this
    layout, cacqr.hpp:174-193): the Gram matrix of an image and Q = B T with either side as an image must equal the column-major calls
    BIT FOR BIT -- the layout changes addresses only, not the order of any sum -- and a plain fp64 torch product to 1e-13."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    n = 256
    torch.manual_seed(m)
    A = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5            # column-major m x n
    A32 = A.T.reshape(m // 32, 32, n).transpose(1, 2).contiguous()              # tile t: [column][32 rows]
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5)
    Tcm = T.T.contiguous()
    G0, G1 = torch.zeros((n, n), dtype=torch.float64, device="cuda"), torch.zeros((n, n), dtype=torch.float64, device="cuda")
    h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), m, 0.0, capi.ptr(G0), n)
    h.call("capi_dsyrk_panel32", n, m, 1.0, capi.ptr(A32), 0.0, capi.ptr(G1), n)
    Q0 = torch.zeros((n, m), dtype=torch.float64, device="cuda")
    h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 0.75, capi.ptr(Tcm), n, capi.ptr(A), m, capi.ptr(Q0), m)
    outs = {}
    for src_t in (False, True):
        for dst_t in (False, True):
            if not (src_t or dst_t):
                continue
            Q = torch.zeros((m // 32, n, 32) if dst_t else (n, m), dtype=torch.float64, device="cuda")
            h.call("capi_dtrmm_right_panel32", m, n, 0.75, capi.ptr(Tcm), n, capi.ptr(A32 if src_t else A), 0 if src_t else m, capi.ptr(Q), 0 if dst_t else m)
            outs[(src_t, dst_t)] = Q.transpose(1, 2).reshape(m, n).T.contiguous() if dst_t else Q
    h.sync()
    assert torch.equal(G0, G1)
    for k, Q in outs.items():
        assert torch.equal(Q, Q0), k
    Am = A.T
    Qref = 0.75 * (Am @ T)
    assert (Q0.T - Qref).abs().max().item() <= 1e-13 * Qref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("n,pad,uplo,trans,alpha,beta", [(512, 0, 1, 0, 1.0, 0.0), (768, 2, 1, 0, -0.5, 0.0), (1024, 0, 1, 0, 1.0, 0.0), (1024, 3, 0, 1, 2.0, 0.0),
                                                          (1000, 0, 1, 0, 1.0, 0.0), (1024, 0, 0, 0, 1.0, 0.0), (512, 2, 1, 1, 1.0, 0.0), (768, 0, 1, 0, 1.0, 0.75),
                                                          (2048, 0, 1, 0, 1.0, 0.0)])
def test_tall_products_wider_than_256(n, pad, uplo, trans, alpha, beta):
    """Tall-skinny products at the widths of BASELINE config 5 (n = 1024) and around it, m just above 64 n so that the tall paths
    engage: Gram matrix by 256-blocks (diagonal blocks on the full-width kernel, off-diagonal blocks as split-K products), right
    TRMM with a clean copy of T, one launch per 256-column block when op(T) is upper; ragged m, padded leading dimension, the
    widths that do NOT divide by 256 (generic path), lower / transposed T, alpha, and beta on the Gram matrix -- all against torch fp64."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    torch.manual_seed(n + pad + 10 * uplo + trans)
    m = 64 * n + 77
    ld = m + pad
    A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5           # column-major m x n, leading dimension ld
    Q = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    Tfull = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5        # logical matrix; only its `uplo` triangle may be read
    Tcm = Tfull.T.contiguous()                                                   # column-major storage
    Tlog = torch.triu(Tfull) if uplo == 1 else torch.tril(Tfull)
    opT = Tlog.T if trans else Tlog
    G0 = torch.rand((n, n), dtype=torch.float64, device="cuda")
    G = G0.clone()
    torch.cuda.synchronize()
    h.call("capi_dsyrk", 1, 1, n, m, alpha, capi.ptr(A), ld, beta, capi.ptr(G), n)
    h.call("capi_dtrmm_oop", 1, uplo, trans, 0, m, n, alpha, capi.ptr(Tcm), n, capi.ptr(A), ld, capi.ptr(Q), ld)
    h.sync()
    Am = A[:, :m].T
    Gref = alpha * torch.triu(Am.T @ Am) + beta * torch.triu(G0.T)
    assert (torch.triu(G.T) - Gref).abs().max().item() <= 1e-13 * Gref.abs().max().item()
    assert torch.equal(torch.tril(G.T, -1), torch.tril(G0.T, -1))                # the other triangle of C is neither read nor written
    Qref = alpha * (Am @ opT)
    assert (Q[:, :m].T - Qref).abs().max().item() <= 1e-13 * Qref.abs().max().item()
    assert torch.count_nonzero(Q[:, m:]).item() == 0
    if n <= 768:                                          # the in-place form (cblas_dtrmm's own) on the same operands: bit-identical result
        B = A.clone()
        torch.cuda.synchronize()
        h.call("capi_dtrmm", 1, uplo, trans, 0, m, n, alpha, capi.ptr(Tcm), n, capi.ptr(B), ld)
        h.sync()
        assert torch.equal(B[:, :m], Q[:, :m])
        assert torch.equal(B[:, m:], A[:, m:])
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("side,uplo,trans,diag", list(itertools.product((0, 1), (0, 1), (0, 1), (0, 1))))
@pytest.mark.parametrize("ntri,nfree", [(4096, 4096), (2048, 8192), (8192, 4096)])
def test_dtrmm_tile_pairs(side, uplo, trans, diag, ntri, nfree):
    """Orders from 4096 up run as tile PAIRS (gemm_f64.hip: dtrmm_pair_kernel: tile b and tile ntri - 1 - b in one workgroup, the
    second walked backwards in k): all sixteen forms against a plain fp64 product of the masked triangle, 1e-12 relative; the
    in-place form must give the same bits as the out-of-place one."""
    import torch
    from capital_amd import capi
    if diag and ntri != 4096:
        pytest.skip("unit diagonal is covered at the square size")
    h = capi.Handle(0)
    torch.manual_seed(side * 8 + uplo * 4 + trans * 2 + diag)
    m, n = (ntri, nfree) if side == 0 else (nfree, ntri)
    Tt = torch.rand((ntri, ntri), dtype=torch.float64, device="cuda") - 0.5     # column-major T = Tt^T, junk in the unreferenced triangle
    Bt = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5           # column-major m x n
    Ct = torch.full((n, m), float("nan"), dtype=torch.float64, device="cuda")   # beta == 0: never read
    Te = torch.tril(Tt) if uplo == 1 else torch.triu(Tt)
    if diag:
        Te = Te - torch.diag(torch.diagonal(Te)) + torch.eye(ntri, dtype=torch.float64, device="cuda")
    opTt = Te if trans == 0 else Te.T                                            # op(T)^T
    ref = -1.5 * (Bt @ opTt if side == 0 else opTt @ Bt)
    torch.cuda.synchronize()
    h.call("capi_dtrmm_oop", side, uplo, trans, diag, m, n, -1.5, capi.ptr(Tt), ntri, capi.ptr(Bt), m, capi.ptr(Ct), m)
    h.sync()
    assert (Ct - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
    h.call("capi_dtrmm", side, uplo, trans, diag, m, n, -1.5, capi.ptr(Tt), ntri, capi.ptr(Bt), m)
    h.sync()
    assert torch.equal(Bt, Ct)


@pytest.mark.gpu
@pytest.mark.parametrize("uplo", (0, 1))
@pytest.mark.parametrize("n,k", [(2304, 512), (4480, 192), (6016, 200)])
def test_triangular_output_in_bands(uplo, n, k):
    """From 16 tile rows up a triangular output's tiles are enumerated in bands of 8 tile rows (gemm_f64.hip: tile_of_dims), also in
    the 64-tile re-cut of a partial last round: every element of the wanted triangle is produced exactly once (beta = 1 on a known C
    would show a tile computed twice), the other triangle is untouched."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    torch.manual_seed(n + uplo)
    A = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5     # column-major k x n
    C0 = torch.rand((n, n), dtype=torch.float64, device="cuda")
    C1 = C0.clone()
    torch.cuda.synchronize()
    h.call("capi_dsyrk", uplo, 1, n, k, -1.0, capi.ptr(A), k, 1.0, capi.ptr(C1), n)      # wanted triangle of C -= A^T A
    h.sync()
    full = C0 - A @ A.T
    # column-major upper triangle == lower triangle of the row-major tensor
    want, keep = (torch.tril, torch.triu) if uplo == 1 else (torch.triu, torch.tril)
    assert (want(C1) - want(full)).abs().max().item() <= 1e-13 * full.abs().max().item()
    off = 1 if uplo == 1 else -1
    assert torch.equal(keep(C1, off), keep(C0, off))


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(5000, 300), (6016, 200)])
def test_partial_last_round_is_recut_into_64_tiles(n, k):
    """Orders whose 128-tiling leaves a partial last round (1600 = 3 x 512 + 64 tiles for gemm 5000, 1128 = 2 x 512 + 104
    for the triangle of 6016): the tail tiles are launched as 64-tiles; results must not change (gemm and triangular
    output, 1e-13 relative)."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    torch.manual_seed(n)
    A = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5     # column-major k x n  (A^T is n x k)
    B = torch.rand((n, k), dtype=torch.float64, device="cuda") - 0.5
    C0 = torch.rand((n, n), dtype=torch.float64, device="cuda")
    C1 = C0.clone()
    h.call("capi_dgemm", 1, 0, n, n, k, -1.0, capi.ptr(A), k, capi.ptr(B), k, 1.0, capi.ptr(C1), n)     # C -= A^T B
    h.sync()
    ref = C0 - (B @ A.T)             # logical C = C0 - At*Bm with At = A (n x k rows), column-major view: C1 tensor holds C^T
    assert (C1 - ref).abs().max().item() <= 1e-13 * ref.abs().max().item()
    C2 = C0.clone()
    h.call("capi_dgemmt", 1, 1, 0, n, k, -1.0, capi.ptr(A), k, capi.ptr(A), k, 1.0, capi.ptr(C2), n)    # upper of C -= A^T A
    h.sync()
    full = C0 - (A @ A.T)
    # column-major upper triangle == lower triangle of the row-major tensor
    assert (torch.tril(C2) - torch.tril(full)).abs().max().item() <= 1e-13 * full.abs().max().item()
    assert torch.equal(torch.triu(C2, 1), torch.triu(C0, 1))             # the other triangle is untouched


@pytest.mark.gpu
def test_randomized_large_shapes_against_torch():
    """40 random cases at the orders where the launch rules change (16+ tile rows: banded triangular order; whole-round TRMMs: tile
    pairs; partial last rounds: the 64-tile re-cut; 64- vs 128-tiles), odd sizes and padded leading dimensions included: syrk / gemmt
    in all four uplo x trans forms and trmm in all eight side x uplo x trans forms against torch fp64, 1e-12 relative; everything
    outside the wanted triangle / beyond the rows of C must stay as it was."""
    import torch
    from capital_amd import capi
    h = capi.Handle(0)
    rng = np.random.default_rng(20261004)
    dev = "cuda"
    for case in range(40):
        kind = case % 2
        if kind == 0:                                            # C(tri) = alpha op(A)^T-ish product + beta C
            n = int(rng.choice([2048, 2304, 2560 + 37, 3000, 4096, 4480, 5000 + 1, 6016]))
            k = int(rng.integers(40, 700))
            uplo, trans = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            pad = int(rng.integers(0, 3)) * 2
            ldc = n + pad
            # column-major A is k x n when trans == 1 (C = A^T A), n x k otherwise (C = A A^T); row-major tensors hold the transposes
            rows, cols = (k, n) if trans == 1 else (n, k)
            lda = rows + pad
            At = torch.rand((cols, lda), dtype=torch.float64, device=dev) - 0.5
            Ct0 = torch.rand((n, ldc), dtype=torch.float64, device=dev)
            Ct = Ct0.clone()
            torch.cuda.synchronize()
            h.call("capi_dsyrk", uplo, trans, n, k, -0.75, capi.ptr(At), lda, 0.5, capi.ptr(Ct), ldc)
            h.sync()
            Am = At[:, :rows].T                                   # the column-major matrix, logically rows x cols
            G = (Am.T @ Am) if trans == 1 else (Am @ Am.T)        # n x n, symmetric
            full = -0.75 * G + 0.5 * Ct0[:, :n].T                 # logical C
            got = Ct[:, :n].T
            want = torch.triu if uplo == 1 else torch.tril
            assert (want(got) - want(full)).abs().max().item() <= 1e-12 * full.abs().max().item(), (case, n, k, uplo, trans)
            other = torch.tril(got, -1) if uplo == 1 else torch.triu(got, 1)
            other0 = torch.tril(Ct0[:, :n].T, -1) if uplo == 1 else torch.triu(Ct0[:, :n].T, 1)
            assert torch.equal(other, other0), (case, "other triangle")
            assert torch.equal(Ct[:, n:], Ct0[:, n:]), (case, "padding rows")
        else:
            ntri = int(rng.choice([2048, 2176, 3072 + 19, 4096, 4096 + 128, 6144, 8192]))
            nfree = int(rng.choice([1024, 2048 + 5, 4096, 8192]))
            side, uplo, trans, diag = (int(rng.integers(0, 2)) for _ in range(4))
            m, n = (ntri, nfree) if side == 0 else (nfree, ntri)
            Tt = torch.rand((ntri, ntri), dtype=torch.float64, device=dev) - 0.5
            Bt = torch.rand((n, m), dtype=torch.float64, device=dev) - 0.5
            Ct = torch.full((n, m), float("nan"), dtype=torch.float64, device=dev)
            Te = torch.tril(Tt) if uplo == 1 else torch.triu(Tt)
            if diag:
                Te = Te - torch.diag(torch.diagonal(Te)) + torch.eye(ntri, dtype=torch.float64, device=dev)
            opTt = Te if trans == 0 else Te.T
            ref = 1.25 * (Bt @ opTt if side == 0 else opTt @ Bt)
            torch.cuda.synchronize()
            h.call("capi_dtrmm_oop", side, uplo, trans, diag, m, n, 1.25, capi.ptr(Tt), ntri, capi.ptr(Bt), m, capi.ptr(Ct), m)
            h.sync()
            assert (Ct - ref).abs().max().item() <= 1e-12 * ref.abs().max().item(), (case, ntri, nfree, side, uplo, trans, diag)


def test_randomized_shapes_against_numpy(hip):
    """120 random cases over every BLAS-level entry point, orders 1..700 (crossing the 32-tile burst kernel, the 64- and
    128-tile kernels, split-K and the ragged edges of each), odd leading dimensions, all flag combinations; numpy fp64 as
    the reference, 1e-13 * k relative to the largest entry.  Rows of C beyond m (ld padding) must stay untouched."""
    from capital_amd import capi
    # CAPITAL_FUZZ_SEED / CAPITAL_FUZZ_CASES widen the sweep for one-off runs (e.g. with CAPI_FORCE_TS=128 CAPI_SMALL=0 in the
    # environment, which pins the kernel choice for the whole process)
    import os
    rng = np.random.default_rng(int(os.environ.get("CAPITAL_FUZZ_SEED", "20261004")))
    dims = lambda hi=700: int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 255, 257, 511, 513, int(rng.integers(1, hi))]))
    for case in range(int(os.environ.get("CAPITAL_FUZZ_CASES", "120"))):
        kind = case % 4
        pad = int(rng.integers(0, 4))
        alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, rng.uniform(-2, 2)]))
        if kind == 0:                                           # gemm
            m, n, k, ta, tb = dims(), dims(), dims(), int(rng.integers(2)), int(rng.integers(2))
            A = _rand(rng, (k if ta else m) + pad, m if ta else k); B = _rand(rng, (n if tb else k) + pad, k if tb else n)
            Cf = _rand(rng, m + pad, n)
            opA = A[:k, :].T if ta else A[:m, :]; opB = B[:n, :].T if tb else B[:k, :]
            ref = alpha * (opA @ opB) + beta * Cf[:m, :]
            dA, dB, dC = capi.to_device(A), capi.to_device(B), capi.to_device(Cf)
            hip.call("capi_dgemm", ta, tb, m, n, k, alpha, capi.ptr(dA), A.shape[0], capi.ptr(dB), B.shape[0], beta, capi.ptr(dC), m + pad)
            got = capi.to_host(dC)
            _check(got[:m, :], ref, k, np.abs(ref).max())
        elif kind == 1:                                         # triangular-output gemm (A^T B or A B^T), one triangle only
            n, k, uplo, ta = dims(), dims(), int(rng.integers(2)), int(rng.integers(2))
            A = _rand(rng, (k if ta else n) + pad, n if ta else k); B = _rand(rng, (k if ta else n) + pad, n if ta else k)
            Cf = _rand(rng, n + pad, n)
            full = alpha * ((A[:k, :].T @ B[:k, :]) if ta else (A[:n, :] @ B[:n, :].T)) + beta * Cf[:n, :]
            dA, dB, dC = capi.to_device(A), capi.to_device(B), capi.to_device(Cf)
            hip.call("capi_dgemmt", uplo, ta, 0 if ta else 1, n, k, alpha, capi.ptr(dA), A.shape[0], capi.ptr(dB), B.shape[0], beta, capi.ptr(dC), n + pad)
            got = capi.to_host(dC)
            tri = np.triu if uplo == 1 else np.tril
            other = (lambda x: np.tril(x, -1)) if uplo == 1 else (lambda x: np.triu(x, 1))
            _check(tri(got[:n, :]), tri(full), k, np.abs(full).max())
            np.testing.assert_array_equal(other(got[:n, :]), other(Cf[:n, :]))
            m = n
        else:                                                   # out-of-place trmm (kind 2: left, 3: right)
            side = 0 if kind == 2 else 1
            m, n = dims(), dims()
            nt = m if side == 0 else n
            uplo, trans, diag = int(rng.integers(2)), int(rng.integers(2)), int(rng.integers(2))
            Tf = _rand(rng, nt + pad, nt); B = _rand(rng, m + pad, n); Cf = _rand(rng, m + pad, n)
            T = np.triu(Tf[:nt, :]) if uplo == 1 else np.tril(Tf[:nt, :])
            if diag == 1: np.fill_diagonal(T, 1.0)
            opT = T.T if trans else T
            ref = alpha * (opT @ B[:m, :] if side == 0 else B[:m, :] @ opT)
            dT, dB, dC = capi.to_device(Tf), capi.to_device(B), capi.to_device(Cf)
            hip.call("capi_dtrmm_oop", side, uplo, trans, diag, m, n, alpha, capi.ptr(dT), nt + pad, capi.ptr(dB), m + pad, capi.ptr(dC), m + pad)
            got = capi.to_host(dC)
            _check(got[:m, :], ref, nt, np.abs(ref).max() + 1.0)
        if pad:
            np.testing.assert_array_equal(got[m:, :], Cf[m:, :])


@pytest.mark.gpu
def test_panel32_entry_points_reject_what_they_cannot_serve(hip):
    """The panel32 forms exist for ONE shape family (n = 256, whole 32-row tiles, tall): everything else must come back as an argument error
    with a message, not run a kernel on a layout it does not understand."""
    import torch
    from capital_amd import capi
    n, m = 256, 32 * 1024
    A = torch.zeros((m // 32, n, 32), dtype=torch.float64, device="cuda")
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    T = torch.eye(n, dtype=torch.float64, device="cuda")
    L = capi.load()
    h = hip.h
    import ctypes as C
    d = C.c_double
    assert L.capi_dsyrk_panel32(h, 128, m, d(1.0), capi.ptr(A), d(0.0), capi.ptr(G), n) != 0            # n != 256
    assert L.capi_dsyrk_panel32(h, n, m + 16, d(1.0), capi.ptr(A), d(0.0), capi.ptr(G), n) != 0         # k not a multiple of 32
    assert L.capi_dsyrk_panel32(h, n, 32 * 100, d(1.0), capi.ptr(A), d(0.0), capi.ptr(G), n) != 0       # not tall (k < 64 n)
    assert L.capi_dsyrk_panel32(h, n, m, d(1.0), capi.ptr(A), d(0.0), capi.ptr(G), n - 1) != 0          # ldc < n
    assert b"panel32" in L.capi_last_error(h) or b"operands" in L.capi_last_error(h)
    assert L.capi_dtrmm_right_panel32(h, m + 8, n, d(1.0), capi.ptr(T), n, capi.ptr(A), 0, capi.ptr(A), 0) != 0     # ragged m, and C aliases B
    assert L.capi_dtrmm_right_panel32(h, m, n, d(1.0), capi.ptr(T), n, capi.ptr(A), 0, capi.ptr(A), 0) != 0         # C aliases B
    assert L.capi_dtrmm_right_panel32(h, m, n, d(1.0), capi.ptr(T), n - 1, capi.ptr(A), 0, capi.ptr(G), 0) != 0     # ldt < n
    # and the rounds switch hands back what was set before
    was = C.c_int(-1)
    assert L.capi_set_launch_rounds(h, 1, C.byref(was)) == 0 and was.value == 0
    assert L.capi_set_launch_rounds(h, 0, C.byref(was)) == 0 and was.value == 1
    assert L.capi_set_launch_rounds(h, 2, None) != 0


@pytest.mark.gpu
def test_launches_in_resident_rounds_give_the_same_products(hip):
    """capi_set_launch_rounds(1) (what cholinv::factor turns on for grids): plain products and triangular outputs one resident round of 512
    tiles per launch, TRMMs as equal-work tile pairs one round per launch.  The tiles, their k order and so every sum are those of the
    one-launch form: dgemm and dsyrk must come out BIT-identical; the pair kernel walks its long tile backwards in k, so the TRMM is held
    to 1e-13 against the default form and a plain torch product."""
    import ctypes as C
    import torch
    from capital_amd import capi
    L = capi.load()
    n = 8192                                                       # 4096 tiles of 128 (8 rounds); triangle 2080 tiles (4 rounds + a re-cut tail); 2048 pair workgroups
    torch.manual_seed(8192)
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5).T.contiguous()      # column-major upper

    def products():
        Cg = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        Cs = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        Ct = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        hip.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, 0.0, capi.ptr(Cg), n)
        hip.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 0.0, capi.ptr(Cs), n)
        hip.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(T), n, capi.ptr(B), n, capi.ptr(Ct), n)
        hip.sync()
        return Cg, Cs, Ct

    base = products()
    was = C.c_int(-1)
    assert L.capi_set_launch_rounds(hip.h, 1, C.byref(was)) == 0
    try:
        rounds = products()
    finally:
        assert L.capi_set_launch_rounds(hip.h, 0, None) == 0
    assert torch.equal(base[0], rounds[0])
    assert torch.equal(base[1], rounds[1])
    scale = base[2].abs().max().item()
    assert (base[2] - rounds[2]).abs().max().item() <= 1e-13 * scale
    ref = (torch.triu(T.T).T @ B.T).T                               # column-major: Ct = T^T B
    assert (rounds[2] - ref).abs().max().item() <= 1e-12 * scale


@pytest.mark.gpu
def test_interval_stamps_of_the_tile_kernels(hip):
    """capi_prof_collect_intervals (round 4): every recorded launch of the tile kernels stamps its own execution interval.  Launches of ONE stream
    run back to back: the union of the intervals equals their sum to within dispatch gaps and stays inside the HIP-event bracket of the whole
    sequence; a product launched in resident rounds records one interval per round, with the same flops in total; variant 100 + o selects both
    128-tile symbols of an orientation (dgemm_tile_kernel and dtrmm_pair_kernel), 8 + v exactly one."""
    import ctypes as C
    import torch
    from capital_amd import capi
    L = capi.load()
    n = 8192
    torch.manual_seed(7)
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5).T.contiguous()
    Cs = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    Ct = torch.zeros((n, n), dtype=torch.float64, device="cuda")

    def collect(code):
        v = [C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()]
        assert L.capi_prof_collect_intervals(hip.h, code, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), C.byref(v[4])) == 0
        return v[0].value, v[1].value, v[2].value, v[3].value, v[4].value

    def run(rounds):
        assert L.capi_set_launch_rounds(hip.h, rounds, None) == 0
        try:
            hip.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 0.0, capi.ptr(Cs), n)           # warm-up (workspaces, LDS limits)
            hip.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(T), n, capi.ptr(A), n, capi.ptr(Ct), n)
            hip.sync()
            assert L.capi_prof_enable(hip.h, 1) == 0
            ms = C.c_float()
            hip.call("capi_timer_start")
            hip.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 0.0, capi.ptr(Cs), n)           # TN, triangular output
            hip.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(T), n, capi.ptr(A), n, capi.ptr(Ct), n)   # left, upper, transposed: TN
            hip.call("capi_timer_stop_ms", C.byref(ms))
            assert L.capi_prof_enable(hip.h, 0) == 0
            return ms.value, collect(103), collect(11), collect(27), collect(-1)
        finally:
            assert L.capi_set_launch_rounds(hip.h, 0, None) == 0

    flops = float(n) * (n + 1.0) * n + float(n) * n * n                     # syrk N (N + 1) K + trmm M^2 N
    for rounds in (0, 1):
        total_ms, both, tile, pair, every = run(rounds)
        launches, union, ssum, fl, mx = both
        # (the 32 tiles of the triangle's partial last round are re-cut into 64-tiles: that launch is in neither the record's flops nor its interval)
        assert flops * (1 - 32.0 / 2080 * 0.51) - 1e-3 * flops <= fl <= flops * (1 + 1e-12), (rounds, fl, flops)
        assert tile[0] + pair[0] == launches and launches >= (2 if rounds == 0 else 8)
        assert 0.0 < union <= ssum * (1 + 1e-9) and mx <= union * (1 + 1e-9)
        assert union <= total_ms * 1.02 and ssum >= 0.85 * total_ms, (rounds, union, ssum, total_ms)      # one stream: back to back, inside the bracket
        assert every[3] >= fl
        assert 40.0 < fl / (union * 1e-3) / 1e12 < 78.6                       # a rate a tile kernel can have


@pytest.mark.gpu
def test_reserved_cus_leave_the_products_unchanged(hip):
    """capi_reserve_cus: the compute stream under a CU mask (round 4: what a caller with latency-critical transfers can turn on so that kernels of
    the communication stream start at once beside a tile launch).  Multiples of 32 only (one CU per shader engine and XCD: anything else leaves
    the engines of an XCD unequal and costs up to 65 % of the tile kernel's rate, profiles/r4_overlap_contention.txt); same bits with and without;
    a borrowed stream cannot be masked.  Own handle: the session's would keep the mask."""
    import ctypes as C
    import torch
    from capital_amd import capi
    L = capi.load()
    assert L.capi_reserve_cus(hip.h, 32) != 0                   # the session's handle borrows torch's stream
    hnd = capi.Handle(0, own_stream=True)
    try:
        n = 4096
        torch.manual_seed(3)
        A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
        torch.cuda.synchronize()                                 # (the handle's own stream is not torch's)
        out = []
        for reserve in (0, 32, 64, 0):
            assert L.capi_reserve_cus(hnd.h, reserve) == 0
            Cc = torch.zeros((n, n), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            assert L.capi_set_launch_rounds(hnd.h, 1, None) == 0
            hnd.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(A), n, capi.ptr(A), n, 0.0, capi.ptr(Cc), n)
            hnd.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 1.0, capi.ptr(Cc), n)
            hnd.sync()
            out.append(Cc)
        for o in out[1:]:
            assert torch.equal(out[0], o)
        assert L.capi_reserve_cus(hnd.h, 8) != 0 and L.capi_reserve_cus(hnd.h, -32) != 0 and L.capi_reserve_cus(hnd.h, 256) != 0
    finally:
        hnd.close()
