"""The RCCL wrappers (capi_comm_*, capi_bcast/allreduce/reduce/allgather/sendrecv_replace; the reference's MPI
collectives C1-C10) against the real librccl.  A 1-GPU box admits exactly one RCCL configuration -- a communicator of
one rank -- so CAPI_RCCL_FORCE routes that through the library instead of the size-1 short cut; multi-rank semantics
are covered by tests/test_multirank_gloo.py on the CPU."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_wrappers_single_rank():
    env = dict(os.environ, CAPI_RCCL_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_single_rank.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
