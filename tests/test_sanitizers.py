"""CPU only: the host-side layer (capital_amd/src: cholinv / summa / cacqr schedules, the arena and its views, topo::square transfer
sets, csrc/pair_paths.h) built with -fsanitize=address,undefined against the oracle-backed shim (tests/cpu_shim: `make asan`) and run
by gloo ranks with libasan preloaded (race detection / sanitizers; GPU sanitizers are not available on the pool).
A heap overrun in a view, a read of freed arena memory, a signed overflow in an index: any report fails the run."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from test_multirank_gloo import SHIM, _free_port

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def asan_lib():
    subprocess.check_call(["make", "-C", SHIM, "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this image")
    return os.path.join(SHIM, "libcapital_driver_cpu_asan.so"), os.path.realpath(libasan)


def _run(asan_lib, world, cfg, timeout=900):
    lib, libasan = asan_lib
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   GLOO_SOCKET_IFNAME="lo", CAPITAL_MIN_CHUNK_COLS="8", CAPITAL_MULTIPATH="2", CAPITAL_MULTIPATH_MIN="8",
                   CAPITAL_SHIM_LIB=lib, LD_PRELOAD=libasan,
                   # python and torch are not instrumented: leaks and the interpreter's own allocator games are not what is looked for
                   ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66:allocator_may_return_null=1:verify_asan_link_order=0",
                   UBSAN_OPTIONS="halt_on_error=1:exitcode=67:print_stacktrace=1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(SHIM, "rank_main.py"), json.dumps(cfg)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    joined = "\n".join(o[-4000:] for o in outs)
    assert all(p.returncode == 0 for p in procs), joined
    assert "AddressSanitizer" not in joined and "runtime error:" not in joined, joined


@pytest.mark.parametrize("world,cfg", [
    (1, {"kind": "cholinv", "n": 160, "c": 1, "bc": -3, "ci": 1, "serialize": True, "policy": 2}),
    (4, {"kind": "cholinv", "n": 97, "c": 1, "bc": -2, "ci": 1, "serialize": True, "policy": 1, "chunks": 3}),       # padding, multi-path rows / columns
    (8, {"kind": "cholinv", "n": 192, "c": 2, "bc": -2, "ci": 1, "serialize": True, "policy": 3, "chunks": 4}),      # 2x2x2: depth halves, pipelines, packed pieces
    (8, {"kind": "cacqr", "m": 1000, "n": 48, "c": 2, "variant": 2, "serialize": True, "ci": 0, "bc": -1, "chunks": 3}),
    (2, {"kind": "cacqr", "m": 4096, "n": 32, "variant": 2, "serialize": True}),
])
def test_host_layer_under_asan_ubsan(asan_lib, world, cfg):
    with tempfile.TemporaryDirectory() as d:
        _run(asan_lib, world, dict(cfg, dir=d))
        z = np.load(os.path.join(d, "rank0.npz"))
        assert float(z["residual"]) <= 1e-14 or float(z["residual"]) == -1.0
