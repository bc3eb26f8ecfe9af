"""CPU: the multi-path pair-transfer algorithm (capital_amd/csrc/pair_paths.h -- the template that comm_rccl.hip instantiates over RCCL)
run by N rank threads over an in-memory transport with RCCL's point-to-point matching rules (tests/pair_paths/pair_paths_sim.cpp, built
with ASan + UBSan).  Delivery must be exact, and -- the property the bandwidth claim of DESIGN.md section 6 rests on -- in each of the two
phases every directed link carries AT MOST ONE message of AT MOST ONE unit (count / nranks, rounded up to even)."""
import json
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_paths")


@pytest.fixture(scope="module")
def sim():
    subprocess.check_call(["make", "-C", HERE, "-s"])
    return os.path.join(HERE, "pair_paths_sim")


def _run(sim, n, count, minc, kind):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([sim, str(n), str(count), str(minc), kind], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n,kind", [(8, "row"), (8, "column"), (8, "depth"), (8, "transpose"), (4, "row"), (4, "column"), (4, "transpose")])
@pytest.mark.parametrize("count", [4096, 1001, 7])
def test_grid_transfer_sets_deliver_and_load_every_link_once(sim, n, kind, count):
    r = _run(sim, n, count, 1, kind)            # threshold 1: even these messages are cut into units and relayed
    assert r["wrong"] == 0 and r["bad_rc"] == 0 and r["leftover"] == 0
    assert r["transfers"] == {"row": n // 2, "column": n // 2, "depth": n, "transpose": n // 2}[kind]
    assert r["max_msgs_per_link_per_phase"] <= 1
    assert r["max_message"] <= r["unit"] == ((count + n - 1) // n + 1) // 2 * 2
    if count >= 2 * n:
        # a source fans out over all its n - 1 links in phase 1; phase 2 uses the n - 1 links INTO every destination
        assert r["links_phase1"] == r["transfers"] * (n - 1) and r["links_phase2"] == r["transfers"] * (n - 1)


@pytest.mark.parametrize("n", [2, 3, 5, 8])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_partial_permutations(sim, n, seed):
    r = _run(sim, n, 999, 1, f"random{seed}")
    assert r["wrong"] == 0 and r["bad_rc"] == 0 and r["leftover"] == 0 and r["max_msgs_per_link_per_phase"] <= 1


def test_short_messages_go_directly(sim):
    r = _run(sim, 8, 512, 1 << 20, "depth")    # below the threshold: one direct message per transfer, no relays
    assert r["wrong"] == 0 and r["links_phase1"] == 8 and r["links_phase2"] == 0 and r["max_message"] == 512
