"""bench.py's host-side helpers (no GPU): the grid / base-case rule per N, the recorded-traffic reader against the committed PMC
passes, and the host-BLAS baseline leg on a tiny sample."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_grid_and_base_case_rule():
    import bench
    for n_gpus, c in bench.GRID_C.items():
        d = int(round((n_gpus // c) ** 0.5))
        assert d * d * c == n_gpus
        bc = bench.bc_mult_for(bench.N_CHOLESKY, d, c, bench.BASE_CASE_ORDER)
        # cholinv.hpp:15-18 restated: t = c d 2^|bc|, bc_loc = n_loc / t, aggregated order = d bc_loc
        n_loc = -(-bench.N_CHOLESKY // d)
        assert d * (n_loc // (c * d * 2 ** (-bc))) == bench.BASE_CASE_ORDER, (n_gpus, bc)
    assert bench.bc_mult_for(32768, 1, 1, 1024) == -5 and bench.bc_mult_for(65536, 1, 1, 1024) == -6
    assert bench.bc_mult_for(65536, 1, 1, bench.BASE_CASE_ORDER_ONE_GPU) == -5 and bench.bc_mult_for(32768, 1, 1, bench.BASE_CASE_ORDER_ONE_GPU) == -4   # one GPU: order 2048
    assert bench.QR_CONFIG5_SLICE[0] * 8 == 1 << 26 and bench.QR_CONFIG5_SLICE[1] == 1024        # N = 8 is BASELINE config 5


def test_recorded_traffic_reads_the_committed_passes(monkeypatch):
    """`roofline.traffic` is a RECORDED figure: the committed PMC passes of the same configuration AND launch form (round 4: n = 65536 on one GPU
    with launches in resident rounds, both symbols of the roofline's kernel) -- nothing else may borrow them."""
    import bench
    monkeypatch.delenv("CAPITAL_NO_LAUNCH_ROUNDS", raising=False)
    t, src = bench.recorded_traffic(65536, 1)
    if os.path.exists(os.path.join(ROOT, "profiles", "r4_pmc_fe_bench_n65536.csv")):
        assert t is not None and 1e9 < t < 2e11 and "recorded" in src and "r4_pmc" in src and "launches" in src
    else:
        assert (t, src) == (None, None)
    assert bench.recorded_traffic(32768, 1) == (None, None)
    assert bench.recorded_traffic(65536, 8) == (None, None)
    monkeypatch.setenv("CAPITAL_NO_LAUNCH_ROUNDS", "1")
    assert bench.recorded_traffic(65536, 1) == (None, None)        # one launch per product is another launch form: no committed pass describes it


def test_host_baseline_leg_runs_and_names_its_library():
    """oracle.host_baseline in its own interpreter on a tiny sample: prints one JSON object with the library it bound (or says
    it fell back to the oracle's own kernels), the thread count, and both schedules' rates"""
    res = subprocess.run([sys.executable, "-m", "oracle.host_baseline", "--n", "512", "--bc", "-2", "--m", "4096", "--qn", "32", "--threads", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    js = json.loads(res.stdout.strip().splitlines()[-1])
    assert js["kind"] in ("host-blas", "port") and js["cores"] >= 1 and js["library"]
    assert js["cholesky"]["tflops"] > 0 and js["cholesky"]["residual"] <= 1e-14
    assert js["cacqr2"]["tflops"] > 0
