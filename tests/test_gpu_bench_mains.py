"""The stand-alone bench mains (capital_amd/bench/*) on the GPU: the reference's own
command lines (bench/cholesky/cholinv.cpp:8-71, bench/qr/cacqr.cpp:8-77, bench/matmult/summa_gemm.cpp), the reference's
output protocol ("total time - <s>" per timed iteration; "m n c bc seconds"), and the residuals its validators print,
which must agree with the CPU oracle's on the same generated input."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, *args):
    path = os.path.join(ROOT, "capital_amd", exe)
    assert os.path.exists(path), f"{path} is not built (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([path, *[str(a) for a in args]], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.parametrize("n,bc", [(1024, 0), (1000, -3)])      # BASELINE config 1 (the reference's CPU-runnable case), and a recursive one
def test_bench_cholinv_cli(oracle, n, bc):
    out = _run("bench_cholinv", n, 1, 0, 1, bc, 0, 0, 2)
    times = [float(x) for x in re.findall(r"^total time - ([0-9.eE+-]+)", out, re.M)]
    assert len(times) == 2 and all(t > 0 for t in times)           # num_iter timed calls after the warm-up
    res = float(re.search(r"^residual - ([0-9.eE+-]+)", out, re.M).group(1))
    A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
    Rref, _, info = oracle.cholinv_factor(A, 0, 1, bc, 1, 1)
    assert info == 0
    assert res <= 1e-14 and abs(res - oracle.cholesky_residual(A, Rref)) <= 5e-16


def test_bench_cholinv_usage():
    path = os.path.join(ROOT, "capital_amd", "bench_cholinv")
    r = subprocess.run([path, "1024"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr


def test_bench_cacqr_cli(oracle):
    m, n = 65536, 256                                              # the reference's example command, section 8c of the survey
    out = _run("bench_cacqr", 2, m, n, 1, 1, 0, 1, 0, 0, 0, 0, 1)
    line = re.search(r"^(\d+) (\d+) (\d+) (-?\d+) ([0-9.eE+-]+)$", out, re.M)
    assert line and (int(line.group(1)), int(line.group(2)), int(line.group(3)), int(line.group(4))) == (m, n, 1, 0)
    assert float(line.group(5)) > 0
    res = float(re.search(r"residual ([0-9.eE+-]+)", out).group(1))
    orth = float(re.search(r"orthogonality ([0-9.eE+-]+)", out).group(1))
    A = oracle.distribute_random(n, m, 0, 0, 1, 1, key=0)
    Q, R, info = oracle.cacqr_factor_1d(A, 1, 2)
    assert info == 0
    assert res <= 1e-14 and orth <= 1e-15
    assert abs(res - oracle.qr_residual(A, Q, R)) <= 5e-16


def test_bench_summa_gemm_cli(oracle, tmp_path, monkeypatch):
    """bench/matmult/summa_gemm.cpp:7-54: same CLI and generators; the product it leaves in C equals A B of the oracle's
    generators elementwise (1e-12 relative), not merely "some time was printed" """
    M, N, K = 512, 384, 640
    monkeypatch.setenv("CAPITAL_BENCH_DUMP", str(tmp_path / "C"))
    out = _run("bench_summa_gemm", M, N, K, 1, 0, 0, 2)
    times = [float(x) for x in re.findall(r"^total time - ([0-9.eE+-]+)", out, re.M)]
    assert len(times) == 2 and all(t > 0 for t in times)
    C = np.fromfile(str(tmp_path / "C.0"), dtype=np.float64).reshape((M, N), order="F")
    A = oracle.distribute_random(K, M, 0, 0, 1, 1, key=0)          # A is M x K, keyed rank / c = 0
    B = oracle.distribute_random(N, K, 0, 0, 1, 1, key=0)          # B keyed -(rank / c) = 0 as well (summa_gemm.cpp:36-37)
    ref = oracle.dgemm(0, 0, 1.0, A, B, 0.0, np.zeros((M, N), order="F"))
    assert np.abs(C - ref).max() <= 1e-12 * np.abs(ref).max()
