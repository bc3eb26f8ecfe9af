"""GPU parity (bit-exact): generators, serialize / pack, triangle removal, block<->cyclic re-indexing against
the oracle's loop-for-loop restatement of structure.hpp:36-129, serialize.hpp:12-150, util.hpp:56-318."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,px,py", [(64, 1, 0, 0), (1000, 1, 0, 0), (1001, 2, 1, 0), (1001, 2, 1, 1), (777, 3, 2, 1), (4096, 2, 0, 1)])
def test_distribute_symmetric_bit_exact(hip, oracle, n, d, px, py):
    from capital_amd import capi
    ref = oracle.distribute_symmetric(n, n, px, py, d, d, key=7)
    dy, dx = ref.shape
    dev = capi.to_device(np.full((dy, dx), np.nan))
    hip.call("capi_distribute_symmetric", capi.ptr(dev), dx, dy, n, n, px, py, d, d, 7, 1)
    np.testing.assert_array_equal(capi.to_host(dev), ref)


@pytest.mark.parametrize("m,n,P,p", [(4096, 64, 1, 0), (100003, 33, 4, 3), (100003, 33, 4, 2), (65536, 256, 8, 5), (31, 7, 1, 0)])
def test_distribute_random_bit_exact(hip, oracle, m, n, P, p):
    """One sequential drand48 stream per rank (structure.hpp:105-129) reproduced with 48-bit LCG jump-ahead."""
    from capital_amd import capi
    key = p  # bench/qr/cacqr.cpp:34: key = rank / c
    ref = oracle.distribute_random(n, m, 0, p, 1, P, key=key)
    dy, dx = ref.shape
    dev = capi.to_device(np.full((dy, dx), np.nan))
    hip.call("capi_distribute_random", capi.ptr(dev), dx, dy, n, m, 0, p, 1, P, key)
    np.testing.assert_array_equal(capi.to_host(dev), ref)


def test_distribute_identity(hip, oracle):
    from capital_amd import capi
    ref = oracle.distribute_identity(101, 101, 1, 1, 2, 2, 3.5)
    dy, dx = ref.shape
    dev = capi.to_device(np.full((dy, dx), np.nan))
    hip.call("capi_distribute_identity", capi.ptr(dev), dx, dy, 101, 101, 1, 1, 2, 2, 3.5)
    np.testing.assert_array_equal(capi.to_host(dev), ref)


def _packed_len(st, n):
    return n * n if st == 0 else n * (n + 1) // 2


@pytest.mark.parametrize("ss,ds", [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0), (2, 2)])
def test_serialize(hip, oracle, ss, ds):
    """The seven specialisations of serialize<S1,S2> (serialize.h:19-70) on a sub-range."""
    import ctypes as C
    from capital_amd import capi
    rng = np.random.default_rng(ss * 3 + ds)
    sn, dn = 90, 70
    src = rng.uniform(size=_packed_len(ss, sn))
    dst0 = rng.uniform(size=_packed_len(ds, dn))
    # diagonal sub-block so that triangular layouts are addressed legally
    ssx, sex, ssy, sey = 20, 60, 20, 60
    dsx, dex, dsy, dey = 10, 50, 10, 50
    ref = dst0.copy()
    dp = C.POINTER(C.c_double)
    oracle.lib().orc_serialize(ss, ds, src.ctypes.data_as(dp), sn, sn, ref.ctypes.data_as(dp), dn, dn, ssx, sex, ssy, sey, dsx, dex, dsy, dey)
    import torch
    dsrc, ddst = torch.from_numpy(src).cuda(), torch.from_numpy(dst0).cuda()
    hip.call("capi_serialize", ss, ds, capi.ptr(dsrc), sn, sn, capi.ptr(ddst), dn, dn, ssx, sex, ssy, sey, dsx, dex, dsy, dey)
    np.testing.assert_array_equal(ddst.cpu().numpy(), ref)


def test_lacpy_trizero_axpby_remove_triangle(hip, oracle):
    from capital_amd import capi
    import torch
    rng = np.random.default_rng(3)
    m, n = 150, 140
    A, B = rng.uniform(size=(m, n)), rng.uniform(size=(m, n))
    for part in (0, 1, 2):
        dA, dB = capi.to_device(A), capi.to_device(B)
        hip.call("capi_dlacpy", part, m, n, capi.ptr(dA), m, capi.ptr(dB), m)
        i, j = np.indices((m, n))
        sel = np.ones((m, n), bool) if part == 0 else (i <= j if part == 1 else i >= j)
        np.testing.assert_array_equal(capi.to_host(dB), np.where(sel, A, B))
    S = rng.uniform(size=(n, n))
    for keep in (0, 1):
        dS = capi.to_device(S)
        hip.call("capi_dtrizero", keep, n, capi.ptr(dS), n)
        np.testing.assert_array_equal(capi.to_host(dS), np.triu(S) if keep else np.tril(S))
    x, y = rng.uniform(size=100001), rng.uniform(size=100001)
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    hip.call("capi_daxpby", x.size, -0.25, capi.ptr(dx), capi.ptr(dy))
    np.testing.assert_array_equal(dy.cpu().numpy(), -0.25 * y + x)       # summa.hpp:33,153
    # util::remove_triangle (util.hpp:266-291) by GLOBAL index on a 2x2 grid piece
    L = rng.uniform(size=(40, 40))
    for d_, (px, py) in ((b'U', (1, 0)), (b'L', (0, 1)), (b'U', (0, 0))):
        dL = capi.to_device(L)
        hip.call("capi_remove_triangle", d_, capi.ptr(dL), 40, 40, px, py, 2)
        jj, ii = np.indices((40, 40))     # L[row j, col i]
        gx, gy = px + ii * 2, py + jj * 2
        ref = np.where((gy > gx) if d_ == b'U' else (gy < gx), 0.0, L)
        np.testing.assert_array_equal(capi.to_host(dL), ref)


@pytest.mark.parametrize("rl,d", [(8, 2), (5, 3), (16, 1)])
def test_block_cyclic(hip, oracle, rl, d):
    import ctypes as C
    import torch
    from capital_amd import capi
    rng = np.random.default_rng(rl + d)
    blocked = rng.uniform(size=rl * rl * d * d)
    ref = np.zeros((rl * d) * (rl * d))
    dp = C.POINTER(C.c_double)
    oracle.lib().orc_block_to_cyclic_rect(blocked.ctypes.data_as(dp), ref.ctypes.data_as(dp), rl, rl, d)
    db, dc = torch.from_numpy(blocked).cuda(), torch.zeros(ref.size, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()             # torch fills on ITS stream; the handle runs on its own
    hip.call("capi_block_to_cyclic", capi.ptr(db), capi.ptr(dc), rl, rl, d)
    hip.sync()
    np.testing.assert_array_equal(dc.cpu().numpy(), ref)
    full = rng.uniform(size=ref.size)
    back = np.zeros_like(blocked)
    oracle.lib().orc_cyclic_to_block_rect(back.ctypes.data_as(dp), full.ctypes.data_as(dp), rl, rl, d)
    db2, dfull = torch.zeros(blocked.size, dtype=torch.float64, device="cuda"), torch.from_numpy(full).cuda()
    torch.cuda.synchronize()
    hip.call("capi_cyclic_to_block", capi.ptr(db2), capi.ptr(dfull), rl, rl, d)
    hip.sync()
    np.testing.assert_array_equal(db2.cpu().numpy(), back)


def test_diff_norms(hip, oracle):
    from capital_amd import capi
    rng = np.random.default_rng(9)
    m, n = 300, 200
    X, Y = rng.uniform(size=(m, n)), rng.uniform(size=(m, n))
    import ctypes as C
    out = (C.c_double * 2)()
    dX, dY = capi.to_device(X), capi.to_device(Y)
    hip.call("capi_diff_norms", 1, m, n, capi.ptr(dX), m, capi.ptr(dY), m, out)
    i, j = np.indices((m, n))
    sel = i <= j
    assert abs(out[0] - ((X - Y)[sel] ** 2).sum()) <= 1e-10 and abs(out[1] - (Y[sel] ** 2).sum()) <= 1e-10


def test_two_streams_and_events(hip, oracle):
    """capi_stream_select / capi_event_record / capi_event_wait: work issued on the communication stream is ordered
    against the compute stream only through events (the chunked SUMMA pipeline relies on exactly this)."""
    import torch
    from capital_amd import capi
    n = 1 << 22
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    z = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        hip.call("capi_daxpby", n, 1.0, capi.ptr(x), capi.ptr(y))          # compute stream: y += x
        hip.call("capi_event_record", 0)
        hip.call("capi_stream_select", 1)
        hip.call("capi_event_wait", 0)
        hip.call("capi_daxpby", n, 0.0, capi.ptr(y), capi.ptr(z))          # comm stream: z = y (must see the update)
        hip.call("capi_event_record", 1)
        hip.call("capi_stream_select", 0)
        hip.call("capi_event_wait", 1)
        hip.call("capi_daxpby", n, 1.0, capi.ptr(z), capi.ptr(y))          # compute stream: y += z
    hip.sync()
    # y: 1 -> 2 | 3 -> 6 | 7 -> 14 ; z follows y before the doubling
    assert float(y[0]) == 14.0 and float(y[-1]) == 14.0 and float(z[12345]) == 7.0


@pytest.mark.parametrize("rl,d", [(8, 2), (5, 3), (16, 1), (33, 2)])
def test_block_cyclic_triangle(hip, oracle, rl, d):
    """packed-triangle pieces (util.hpp:57-102,167-201): device kernels bit-exact against the oracle's restatement, and a round
    trip through the aggregate restores every packed entry that lies in the aggregate's upper triangle"""
    import ctypes as C
    import torch
    from capital_amd import capi
    psz = rl * (rl + 1) // 2
    rng = np.random.default_rng(rl * 7 + d)
    blocked = rng.uniform(0.5, 1.5, size=psz * d * d)
    ref = np.zeros((rl * d) * (rl * d))
    dp = C.POINTER(C.c_double)
    oracle.lib().orc_block_to_cyclic_triangle(blocked.ctypes.data_as(dp), ref.ctypes.data_as(dp), blocked.size, rl, rl, d)
    db = torch.from_numpy(blocked).cuda()
    dc = torch.full((ref.size,), -3.0, dtype=torch.float64, device="cuda")          # every entry of the aggregate must be written
    torch.cuda.synchronize()             # torch fills on ITS stream; the handle runs on its own
    hip.call("capi_block_to_cyclic_tri", capi.ptr(db), capi.ptr(dc), rl, d)
    hip.sync()
    np.testing.assert_array_equal(dc.cpu().numpy(), ref)
    back = np.full(blocked.size, -5.0)
    oracle.lib().orc_cyclic_to_block_triangle(back.ctypes.data_as(dp), ref.ctypes.data_as(dp), blocked.size, rl, rl, d)
    db2 = torch.full((blocked.size,), -9.0, dtype=torch.float64, device="cuda")      # every packed entry must be written
    torch.cuda.synchronize()
    hip.call("capi_cyclic_to_block_tri", capi.ptr(db2), capi.ptr(dc), rl, d)
    hip.sync()
    got = db2.cpu().numpy()
    np.testing.assert_array_equal(got, back)
    changed = np.nonzero(got != blocked)[0]
    assert len(changed) == rl * (d * (d - 1) // 2) and np.all(got[changed] == 0.0)   # local diagonals of the pieces with y > x


@pytest.mark.parametrize("L,d", [(8, 2), (5, 3), (33, 2), (16, 1)])
def test_cyclic_to_local(hip, oracle, L, d):
    """util::cyclic_to_local (util.hpp:131-164) for every slice rank: bit-exact against the oracle's front-to-back restatement"""
    import ctypes as C
    import torch
    from capital_amd import capi
    bc = L * d
    rng = np.random.default_rng(L + 13 * d)
    dp = C.POINTER(C.c_double)
    for sr in range(d * d):
        T, TI = rng.uniform(size=bc * bc), rng.uniform(size=bc * bc)
        rT, rTI = T.copy(), TI.copy()
        oracle.lib().orc_cyclic_to_local.argtypes = [dp, dp] + [C.c_int64] * 4
        oracle.lib().orc_cyclic_to_local(rT.ctypes.data_as(dp), rTI.ctypes.data_as(dp), L, bc, d, sr)
        dT, dTI = torch.from_numpy(T).cuda(), torch.from_numpy(TI).cuda()
        torch.cuda.synchronize()
        hip.call("capi_cyclic_to_local", capi.ptr(dT), capi.ptr(dTI), L, bc, d, sr)
        hip.sync()
        got, goti = dT.cpu().numpy().reshape(bc, bc), dTI.cpu().numpy().reshape(bc, bc)
        np.testing.assert_array_equal(got[:L, :L], rT.reshape(bc, bc)[:L, :L])           # the leading corner (column index first)
        np.testing.assert_array_equal(goti[:L, :L], rTI.reshape(bc, bc)[:L, :L])


@pytest.mark.parametrize("m,n,lda,ldb,off", [(1, 1, 1, 1, 0), (7, 5, 9, 11, 0), (512, 8, 512, 512, 0), (513, 9, 513, 514, 0), (1024, 33, 1030, 1026, 1),
                                             (8192, 8, 8192, 8192, 0), (8192, 9, 8192, 8192, 0), (100003, 3, 100003, 100003, 0), (4096, 70001 // 4096 + 1, 4096, 4100, 0)])
def test_own_copy_kernel_bit_exact(hip, m, n, lda, ldb, off):
    """capi_dlacpy(part 0) and capi_memcpy_d2d_async run on the product's own copy kernel (round 4; no hipMemcpy2DAsync / hipMemcpyAsync blits):
    ragged heights, odd leading dimensions, an operand 8 bytes off a 16-byte boundary, contiguous blocks re-cut into 8192-columns with a rest.
    Everything outside the destination block must stay untouched."""
    import torch
    from capital_amd import capi
    torch.manual_seed(m * 31 + n)
    A = torch.rand(lda * n + off + 4, dtype=torch.float64, device="cuda")
    B = torch.full((ldb * n + off + 4,), -7.0, dtype=torch.float64, device="cuda")
    hip.call("capi_dlacpy", 0, m, n, capi.ptr(A) + 8 * off, lda, capi.ptr(B) + 8 * off, ldb)
    hip.sync()
    ref = torch.full_like(B, -7.0)
    Av = A[off:off + lda * n].view(n, lda)
    ref[off:off + ldb * n].view(n, ldb)[:, :m] = Av[:, :m]
    assert torch.equal(B, ref)
    # the contiguous 1-D form
    cnt = lda * n
    D = torch.full((cnt + 6,), 3.0, dtype=torch.float64, device="cuda")
    hip.call("capi_memcpy_d2d_async", capi.ptr(D) + 8 * (1 + off), capi.ptr(A) + 8 * off, 8 * cnt)
    hip.sync()
    ref = torch.full_like(D, 3.0)
    ref[1 + off:1 + off + cnt] = A[off:off + cnt]
    assert torch.equal(D, ref)
