"""The drop-in boundary as the reference defines it: blas::engine::_gemm/_trmm/_syrk and lapack::engine::_potrf/_trtri/_geqrf/_orgqr
(capital_amd/src/blas/engine.h, src/lapack/engine.h), called by a compiled C++ program (tests/engine_abi/engine_abi.cpp) with
ArgPack_* objects built exactly as the reference's call sites build them (summa.hpp:28,64,139-145; cholinv/policy.h:196-201;
cacqr.hpp:7-29).  Its outputs are compared with the CPU oracle on the same generator inputs.  fp64 tolerance: 1e-12 relative."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "engine_abi")


def _read(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            tag = f.read(32)
            if not tag:
                break
            (count,) = struct.unpack("<q", f.read(8))
            out[tag.split(b"\0")[0].decode()] = np.frombuffer(f.read(8 * count), dtype=np.float64).copy()
    return out


def _close(got, ref, scale=None):
    scale = np.abs(ref).max() if scale is None else scale
    assert np.abs(got - ref).max() <= 1e-12 * max(scale, 1e-300), np.abs(got - ref).max() / scale


def test_engine_call_sites_match_the_oracle(oracle, tmp_path):
    subprocess.check_call(["make", "-C", DIR, "-s"])
    out = str(tmp_path / "engine.bin")
    res = subprocess.run([os.path.join(DIR, "engine_abi"), out], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "engine_abi ok" in res.stdout, res.stdout + res.stderr
    r = _read(out)
    O = oracle
    rnd = lambda rows, cols, key: O.distribute_random(cols, rows, 0, 0, 1, 1, key=key)   # noqa: E731
    spd = lambda n: O.distribute_symmetric(n, n, 0, 0, 1, 1)                              # noqa: E731
    F = lambda a, rows, cols: a.reshape((rows, cols), order="F")                          # noqa: E731

    M, N, K = 200, 136, 168                                          # summa.hpp:28-30
    _close(F(r["gemm_nn"], M, N), 1.5 * rnd(M, K, 1) @ rnd(K, N, 2))
    N, K = 264, 152                                                  # summa.hpp:139-145, both branches
    _close(F(r["gemm_tn"], N, N), -(rnd(K, N, 4).T @ rnd(K, N, 3)))
    _close(F(r["gemm_nt"], N, N), -(rnd(N, K, 5) @ rnd(N, K, 6).T))
    M, N = 192, 120                                                  # summa.hpp:64 / cholinv.hpp:118,150-154
    T, B = np.triu(spd(M)), rnd(M, N, 7)
    step1 = T.T @ B
    _close(F(r["trmm_lut"], M, N), step1)
    _close(F(r["trmm_chain"], M, N), -(T @ step1) @ np.triu(spd(N)))
    agg, span = 160, 150                                             # cholinv/policy.h:196-201
    D = spd(agg)
    ref = D.copy(order="F")
    sub = np.asfortranarray(ref[:span, :span])
    assert O.dpotrf(1, sub) == 0
    got = F(r["bc_potrf"], agg, agg)
    _close(np.triu(got[:span, :span]), np.triu(sub))
    np.testing.assert_array_equal(got[span:, :], D[span:, :])        # outside the span nothing is touched
    np.testing.assert_array_equal(got[:span, span:], D[:span, span:])
    inv = np.asfortranarray(np.triu(sub))
    assert O.dtrtri(1, 0, inv) == 0
    _close(np.triu(F(r["bc_trtri"], agg, agg)[:span, :span]), np.triu(inv))
    assert r["bc_info"][0] == 0.0
    m, n = 3000, 96                                                  # cacqr.hpp:7-29
    A = rnd(m, n, 0)
    Qref, Rref, info = O.cacqr_factor_1d(A, 1, 1)
    assert info == 0
    _close(np.triu(F(r["sweep_gram"], n, n)), np.triu(A.T @ A))
    _close(np.triu(F(r["sweep_R"], n, n)), np.triu(Rref))
    _close(F(r["sweep_Q"], m, n), Qref, scale=1.0)
    m, n = 700, 48                                                   # lapack/interface.hpp:60-88
    A = rnd(m, n, 9)
    Aq = A.copy(order="F")
    tau = O.dgeqrf(Aq)
    _close(F(r["geqrf_A"], m, n), Aq)
    _close(r["geqrf_tau"], tau)
    Qh = F(r["orgqr_Q"], m, n)
    assert np.abs(Qh.T @ Qh - np.eye(n)).max() <= 1e-13
    _close(Qh @ np.triu(Aq[:n, :]), A)
