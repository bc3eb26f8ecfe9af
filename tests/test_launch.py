"""capital_amd/launch.py -- the `mpiexec -n P` of this build (reference: bench/cholesky/cholinv.cpp:8-13 is started as
`mpiexec -n P ./cholinv ...`): `python bench.py --gpus N` without a launcher around it starts its own N ranks.  CPU only: the ranks
here are the gloo / CPU-shim rehearsal ranks of tests/cpu_shim, started through the very same launcher."""
import io
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "cpu_shim")
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def shim_lib():
    subprocess.check_call(["make", "-C", SHIM, "-s"])


def _run(nranks, mode, timeout_s=300.0):
    from capital_amd import launch
    with tempfile.TemporaryFile("w+") as out, tempfile.TemporaryFile("w+") as err:
        t0 = time.monotonic()
        rc = launch.run_ranks(nranks, [sys.executable, os.path.join(SHIM, "bench_rank_main.py"), mode], timeout_s=timeout_s, out=out, err=err,
                              extra_env={"OMP_NUM_THREADS": "1"})
        dt = time.monotonic() - t0
        out.seek(0)
        err.seek(0)
        return rc, out.read(), err.read(), dt


@pytest.mark.parametrize("nranks", [2, 4])
def test_self_launch_relays_one_json_line(shim_lib, nranks):
    rc, out, err, _ = _run(nranks, "ok")
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]    # (gloo's own connection banner goes to stdout)
    assert len(lines) == 1, out                       # rank 0's line and nothing else on stdout
    js = json.loads(lines[0])
    assert js["n_gpus"] == nranks and js["residual"] <= 1e-14 and js["ms_per_step"] > 0
    for r in range(nranks):
        assert f"rank {r} done" in err                # the other ranks' prints went to stderr


def test_failing_rank_ends_the_job(shim_lib):
    rc, out, err, dt = _run(3, "fail")
    assert rc == 7 and out.strip() == "" and "rank 2 exited with 7" in err
    assert dt < 30, dt                                # the sleeping ranks were ended, not waited for


def test_hung_ranks_are_ended_at_the_limit(shim_lib):
    rc, out, err, dt = _run(2, "hang", timeout_s=3.0)
    assert rc == 124 and dt < 30 and "still running" in err


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 2` where fewer than 2 devices exist: one line, non-zero, no traceback, and quickly
    (the first `import torch` of a fresh container is the only slow part)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and time.monotonic() - t0 < 60
    assert "needs 2 GPUs" in res.stderr and "Traceback" not in res.stderr and res.stdout.strip() == ""


def test_bench_rejects_a_world_size_that_disagrees():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=4" in res.stderr and "Traceback" not in res.stderr
