"""CPU: the ring bookkeeping of the asynchronous loopback transport (tests/rccl_loopback/ring_place.h).  The transport is test infrastructure, but the
N > 1 rehearsals on the GPU are only as good as it is: its first version lost a flow-control wait after a wrap (a wrapped message could be written
over unconsumed ones that stood behind an older leftover in the list) -- found on the GPU by the eight-rank rehearsal as an intermittent wrong
factor, fixed, and pinned here by a property test that the old rule fails."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_ring_never_hands_out_an_unconsumed_region(tmp_path):
    exe = tmp_path / "ring_place_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(HERE, "rccl_loopback"), os.path.join(HERE, "rccl_loopback", "ring_place_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), out.stdout + out.stderr
    waited = int(out.stdout.split("checked,")[1].split()[0])
    assert waited > 1000, out.stdout                                         # the lagging receiver did pace the sender: the rule was exercised
