"""N > 1 on hardware: the GPU twin of tests/test_multirank_gloo.py.  N fresh child processes, one per GPU (the parent makes no
GPU call for them: `torch.cuda.device_count()` does not initialise the runtime on this image), each running the PRODUCT --
RCCL world communicator from a shipped unique id, topo::square / topo::rect splits (topology.h:67-143), SUMMA broadcasts and
depth all-reduces (summa.hpp:163-254), base-case gathers (policy.h:160-514), the partner exchange (util.hpp:232-247), the
chunked pipeline on a second stream, the CQR2 Gram all-reduce (cacqr/policy.h:18-24).  The assembled factors must equal the
1-rank oracle: R of an SPD matrix is unique.  Skips (per case) when the box has fewer GPUs than the case needs; the one-rank
case runs everywhere and sends even a 1-rank communicator through RCCL (CAPI_RCCL_FORCE), so the script itself is always exercised.

The `loopback` cases run the same ranks, 2 (1x1x2) and 4 (2x2x1) of them, all on GPU 0, with tests/rccl_loopback standing in for
librccl.so (RCCL proper refuses two ranks on one device): the product's kernels, packed wire formats, d > 1 base cases, K-class SUMMA
and stream / event discipline then execute on a real MI355X on every 1-GPU box; only the transport is not RCCL's.  The `_async` cases
use its asynchronous mode (tests/rccl_loopback/loopback_async.hip): calls only enqueue device work on the caller's stream, as RCCL's
do, so the consumer-side waits of the chunk pipelines (summa.h `P.wait`, cholinv.h EV_BC_*) are really needed -- and
test_async_transport_catches_a_missing_consumer_wait shows that one dropped wait turns the parity check red there (and only there)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpus():
    import torch
    return torch.cuda.device_count()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LOOPBACK = os.path.join(HERE, "rccl_loopback", "librccl_loopback.so")


def _launch(world, cfg, timeout=int(os.environ.get("CAPITAL_TEST_RANK_TIMEOUT_S", "900")), loopback=False, extra_env=None, check=True):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0" if loopback else str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0", CAPITAL_MIN_CHUNK_COLS="64")
        if world == 1:
            env["CAPI_RCCL_FORCE"] = "1"
        if loopback:
            env["CAPI_RCCL_LIB"] = LOOPBACK
            env["CAPI_LOOPBACK_MODE"] = "async" if loopback == "async" else "host"
            env.setdefault("CAPI_LOOPBACK_TIMEOUT_S", "90")
        if extra_env:
            env.update(extra_env)
        logdir = os.environ.get("CAPITAL_TEST_RANK_LOG_DIR")        # diagnostics: every rank's output into a file that survives a killed run
        out = open(os.path.join(logdir, f"rank{r}_of{world}{('_loopback_' + str(loopback)) if loopback else ''}.log"), "w") if logdir else subprocess.PIPE
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_gpu_rank_main.py"), json.dumps(cfg)], env=env,
                                      stdout=out, stderr=subprocess.STDOUT, text=True))
    outs = []
    timed_out = False
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o)
    except subprocess.TimeoutExpired:
        timed_out = True
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if timed_out:                                       # what every rank had printed when the limit struck (each case announces itself)
        outs = [p.communicate()[0] or "" for p in procs]
    outs = [o or "" for o in outs]
    if not check:
        return (not timed_out) and all(p.returncode == 0 for p in procs), outs
    assert not timed_out and all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return True, outs


# N -> c of the d x d x c grid: 1 = 1x1x1, 2 = 1x1x2 (K-slicing), 4 = 2x2x1 (two K-classes per layer), 8 = 2x2x2 (the reference's cubic case)
GRID_C = {1: 1, 2: 2, 4: 1, 8: 2}


@pytest.mark.parametrize("world,loopback", [(1, False), (2, False), (4, False), (8, False), (2, "host"), (4, "host"), (2, "async"), (4, "async")],
                         ids=["rccl1", "rccl2", "rccl4", "rccl8", "loopback2", "loopback4", "loopback2_async", "loopback4_async"])
def test_cholinv_and_cacqr2_on_rccl(oracle, world, loopback):
    if loopback:
        subprocess.check_call(["make", "-C", os.path.dirname(LOOPBACK), "-s"])
    elif _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpus()}")
    c = GRID_C[world]
    n, m_loc, nq = 4096, 1 << 15, 256
    cases = [
        {"tag": "ch_p0", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0},
        {"tag": "ch_p2", "kind": "cholinv", "n": n + 40, "c": c, "bc": -3, "ci": 0, "serialize": False, "policy": 2},      # padded blocks
        {"tag": "ch_p3", "kind": "cholinv", "n": n, "c": c, "bc": -2, "ci": 1, "serialize": True, "policy": 3},
        {"tag": "ch_p1", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": False, "policy": 1},
        {"tag": "ch_chunks", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0, "chunks": 3},
        # multi-path pair transfers (csrc/pair_paths.h) with a low threshold, so that broadcasts, exchanges and depth halves of these
        # orders are cut into units and relayed; "=2" also runs the depth-pair algorithm on the 2-rank grid; then the chunk pipeline on top
        {"tag": "ch_mp", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0,
         "env": {"CAPITAL_MULTIPATH": "2", "CAPITAL_MULTIPATH_MIN": "4096"}},
        {"tag": "ch_mp_chunks", "kind": "cholinv", "n": n + 40, "c": c, "bc": -3, "ci": 1, "serialize": False, "policy": 2, "chunks": 4,
         "env": {"CAPITAL_MULTIPATH": "2", "CAPITAL_MULTIPATH_MIN": "4096"}},
        {"tag": "ch_plain_chunks", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0, "chunks": 3,
         "env": {"CAPITAL_MULTIPATH": "0"}},
        # TRSM mode on the grid (cholinv.h: potrf_rec_grid): assembled R11 / A12, per-rank column ranges through the block TRSM, in-place all-gather
        {"tag": "ch_trsm", "kind": "cholinv", "n": n + 40, "c": c, "bc": -3, "ci": 0, "serialize": True, "policy": 0, "trsm": True},
        {"tag": "qr", "kind": "cacqr", "m": m_loc * world, "n": nq, "serialize": True},
    ]
    if world == 8:
        cases.append({"tag": "qr3d", "kind": "cacqr", "m": 1 << 14, "n": 512, "c": 2, "ci": 1, "bc": -1, "serialize": False})
        cases.append({"tag": "qr3d_chunks", "kind": "cacqr", "m": 1 << 14, "n": 512, "c": 2, "ci": 0, "bc": -1, "serialize": False, "chunks": 3,
                      "env": {"CAPITAL_MULTIPATH_MIN": "4096"}})
        cases.append({"tag": "ch_l1", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": False, "policy": 0, "layout": 1})
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"dir": d, "cases": cases}, loopback=loopback)
        _check_cases(oracle, d, cases, world, c)


def _check_cases(oracle, d, cases, world, c):
        for case in cases:
            tag = case["tag"]
            if case["kind"] == "cholinv":
                nn = case["n"]
                Rg, Ig = np.zeros((nn, nn), order="F"), np.zeros((nn, nn), order="F")
                by_xy = {}
                for r in range(world):
                    z = np.load(os.path.join(d, f"{tag}_rank{r}.npz"))
                    x, y, zz, dd, cc = [int(v) for v in z["xyz"]]
                    assert dd * dd * cc == world and cc == c
                    assert float(z["residual"]) <= 1e-14, (tag, r, float(z["residual"]))
                    by_xy.setdefault((x, y), []).append((z["R"], z["Rinv"]))
                    if zz == 0:
                        oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, dd, dd)
                        if not case.get("trsm", False):
                            oracle.cyclic_insert(Ig, np.asfortranarray(z["Rinv"]), x, y, dd, dd)
                for reps in by_xy.values():                       # depth replicas hold the same blocks
                    for other in reps[1:]:
                        assert np.abs(other[0] - reps[0][0]).max() <= 1e-13 * np.abs(reps[0][0]).max()
                A = oracle.distribute_symmetric(nn, nn, 0, 0, 1, 1)
                Rref, Iref, info = oracle.cholinv_factor(A, case["ci"], 1, case["bc"], c, dd)
                assert info == 0
                assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max(), tag
                if not case.get("trsm", False):
                    assert np.abs(Ig - Iref).max() <= 1e-12 * np.abs(Iref).max(), tag
                assert np.count_nonzero(np.tril(Rg, -1)) == 0
            elif case.get("c", 1) == 1:
                m, nq_ = case["m"], case["n"]
                Ag, Qg = np.zeros((m, nq_), order="F"), np.zeros((m, nq_), order="F")
                Rs = []
                for r in range(world):
                    z = np.load(os.path.join(d, f"{tag}_rank{r}.npz"))
                    np.testing.assert_array_equal(z["A"], oracle.distribute_random(nq_, m, 0, r, 1, world, key=r))
                    oracle.cyclic_insert(Ag, np.asfortranarray(z["A"]), 0, r, 1, world)
                    oracle.cyclic_insert(Qg, np.asfortranarray(z["Q"]), 0, r, 1, world)
                    Rs.append(z["R"])
                    assert float(z["residual"]) <= 1e-14 and float(z["orth"]) <= 1e-15
                for Rr in Rs[1:]:
                    np.testing.assert_array_equal(Rr, Rs[0])      # R is replicated: same all-reduced Gram, same replicated factorisation
                Qref, Rref, info = oracle.cacqr_factor_1d(Ag, world, 2)
                assert info == 0
                assert np.abs(Rs[0] - Rref).max() <= 1e-12 * np.abs(Rref).max()
                assert np.abs(Qg - Qref).max() <= 1e-12
            else:
                m, nq_, cc, dd = case["m"], case["n"], 2, 2
                Ag, Qg, Rg = np.zeros((m, nq_), order="F"), np.zeros((m, nq_), order="F"), np.zeros((nq_, nq_), order="F")
                for r in range(world):
                    z = np.load(os.path.join(d, f"{tag}_rank{r}.npz"))
                    x, y, zz = (r % (cc * cc)) // cc, r // (cc * cc), r % cc            # topology.h:46-50
                    if zz == 0:
                        oracle.cyclic_insert(Ag, np.asfortranarray(z["A"]), x, y, cc, dd)
                        oracle.cyclic_insert(Qg, np.asfortranarray(z["Q"]), x, y, cc, dd)
                        oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, cc, cc)
                Qref, Rref, info = oracle.cacqr_factor_1d(Ag, 1, 2)
                assert info == 0
                assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
                assert np.abs(Qg - Qref).max() <= 1e-12


def _build_driver_without_consumer_wait(tmp):
    """A copy of the host layer in which ONE line is gone -- the compute stream's wait for chunk j's broadcast in every chunk pipeline of
    summa.h (`P.wait(EB + j);`, the consumer side of `P.rec(EB + j)`) -- compiled into a driver library of its own."""
    import shutil
    root = os.path.dirname(HERE)
    for sub in ("src", "test", "drivers"):
        shutil.copytree(os.path.join(root, "capital_amd", sub), os.path.join(tmp, sub))
    summa = os.path.join(tmp, "src", "alg", "matmult", "summa", "summa.h")
    text = open(summa).read()
    assert text.count("P.wait(EB + j);") >= 3
    open(summa, "w").write(text.replace("P.wait(EB + j);", "/* consumer wait removed by the negative test */"))
    out = os.path.join(tmp, "libcapital_driver_nowait.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(root, "include"), os.path.join(tmp, "drivers", "capital_driver.cpp"),
                           "-o", out, "-L" + os.path.join(root, "capital_amd"), "-lcapital_hip", "-Wl,-rpath," + os.path.join(root, "capital_amd")])
    return out


def test_async_transport_catches_a_missing_consumer_wait(oracle):
    """The point of the asynchronous transport: a missing consumer-side wait must FAIL before real RCCL finds it.  The chunked Cholesky
    of the parity cases runs on a driver whose chunk pipelines lack `P.wait(EB + j)` (the product's kernels and library otherwise):
    over the host-staged transport, where every call has completed when it returns, the factors still match the oracle -- that
    transport cannot see the bug; over the asynchronous one (receive-side copies delayed by 2 ms so the window is not a matter of luck)
    they do not."""
    subprocess.check_call(["make", "-C", os.path.dirname(LOOPBACK), "-s"])
    world, c, n = 4, 1, 4096            # 2 x 2 x 1: row / column broadcasts feed every chunk
    cases = [{"tag": "neg_chunks", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0, "chunks": 3,
              "env": {"CAPITAL_MULTIPATH": "0"}}]
    with tempfile.TemporaryDirectory() as build:
        lib = _build_driver_without_consumer_wait(build)
        env = {"CAPITAL_DRIVER_LIB": lib}
        with tempfile.TemporaryDirectory() as d:
            _launch(world, {"dir": d, "cases": cases}, loopback="host", extra_env=env)
            _check_cases(oracle, d, cases, world, c)                  # the synchronous stand-in is blind to it
        with tempfile.TemporaryDirectory() as d:
            ok, outs = _launch(world, {"dir": d, "cases": cases}, loopback="async",
                               extra_env=dict(env, CAPI_LOOPBACK_DELAY_US="2000"), check=False)
            caught = not ok
            if ok:
                try:
                    _check_cases(oracle, d, cases, world, c)
                except AssertionError:
                    caught = True
            assert caught, "the asynchronous transport did not expose the missing wait:\n" + "\n".join(o[-1500:] for o in outs)


THREAD_DRIVER = os.path.join(HERE, "thread_ranks", "libcapital_driver_threads.so")


def _launch_thread_ranks(nproc, threads, cfg, mode, timeout=int(os.environ.get("CAPITAL_TEST_RANK_TIMEOUT_S", "900"))):
    procs = []
    for p in range(nproc):
        env = dict(os.environ, PROC_INDEX=str(p), NPROC=str(nproc), THREADS_PER_PROC=str(threads), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   CAPITAL_MIN_CHUNK_COLS="64", CAPI_RCCL_LIB=LOOPBACK, CAPI_LOOPBACK_MODE=mode, CAPITAL_DRIVER_LIB=THREAD_DRIVER)
        # Every HIP stream of every rank needs a hardware queue of its OWN: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES
        # (default 4) queues, and two ranks whose streams share a queue deadlock in the asynchronous mode -- rank A's device-side wait for rank B's
        # message blocks the very queue B's copy kernel sits in (measured: 2 processes x 4 threads stall in the first case; host-staged mode passes)
        env.setdefault("GPU_MAX_HW_QUEUES", "16")
        # ... and nobody may drain the DEVICE: hipFree does (it waits for every stream of the process), so a rank that frees a block while its sibling's
        # stream sits in a device-side wait for a message the freeing rank has not enqueued yet never returns (measured: the second case stalls in
        # the validator's frees).  The product defers its frees to capi_destroy, the transport leaves its rings to the end of the process.
        env.setdefault("CAPI_DEFER_FREE", "1")
        env.setdefault("CAPI_LOOPBACK_NO_FREE", "1")
        env.setdefault("CAPI_LOOPBACK_TIMEOUT_S", "90")
        logdir = os.environ.get("CAPITAL_TEST_RANK_LOG_DIR")
        out = open(os.path.join(logdir, f"threads_proc{p}_of{nproc}_{mode}.log"), "w") if logdir else subprocess.PIPE
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "_gpu_thread_ranks_main.py"), json.dumps(cfg)], env=env,
                                      stdout=out, stderr=subprocess.STDOUT, text=True))
    outs, timed_out = [], False
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o or "")
    except subprocess.TimeoutExpired:
        timed_out = True
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if timed_out:
        outs = [p.communicate()[0] or "" for p in procs]
    assert not timed_out and all(p.returncode == 0 for p in procs), \
        f"exit codes {[p.returncode for p in procs]}{' (timed out)' if timed_out else ''}\n" + "\n".join(o[-3000:] for o in outs)


@pytest.mark.parametrize("mode", ["host", "async"])
def test_eight_ranks_as_threads_2x2x2_grid_and_cacqr_3d(oracle, mode):
    """BASELINE config 4's own grid, 2 x 2 x 2, on a REAL GPU: eight ranks as threads of four processes (tests/thread_ranks: the host layer with its
    per-rank state thread-local; the pool allows six GPU processes, eight one-rank processes are out of reach), all on GPU 0 over the loopback
    transport in both of its modes.  What had only run on gloo before: the cubic grid's SUMMA with depth all-reduces, the multi-path transfer
    sets at P = 8 (every message cut into eight units, six of them relayed), rank layout 1, TRSM mode on the cube, and the 3-D CholeskyQR2
    (cacqr.hpp:31-170: Gram by broadcast + product + reduce + broadcast, the distributed cholinv on the c x c x c cube, SUMMA right-TRMM) with
    and without the chunk pipeline.  Factors against the 1-rank oracle, as in the one-rank-per-process cases."""
    subprocess.check_call(["make", "-C", os.path.dirname(LOOPBACK), "-s"])
    subprocess.check_call(["make", "-C", os.path.dirname(THREAD_DRIVER), "-s"])
    import torch
    torch.cuda.empty_cache()          # (the session's earlier tests may have left tens of GiB in this process's caching allocator: the eight ranks share the card)
    world, c, n = 8, 2, 4096
    cases = [
        {"tag": "ch_p0", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0},
        {"tag": "ch_p3", "kind": "cholinv", "n": n + 40, "c": c, "bc": -2, "ci": 0, "serialize": False, "policy": 3},
        {"tag": "ch_chunks", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0, "chunks": 3},
        {"tag": "ch_mp", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": True, "policy": 0,
         "env": {"CAPITAL_MULTIPATH": "1", "CAPITAL_MULTIPATH_MIN": "4096"}},
        {"tag": "ch_mp_chunks", "kind": "cholinv", "n": n + 40, "c": c, "bc": -3, "ci": 1, "serialize": False, "policy": 2, "chunks": 4,
         "env": {"CAPITAL_MULTIPATH": "1", "CAPITAL_MULTIPATH_MIN": "4096"}},
        {"tag": "ch_trsm", "kind": "cholinv", "n": n + 40, "c": c, "bc": -3, "ci": 0, "serialize": True, "policy": 0, "trsm": True},
        {"tag": "ch_l1", "kind": "cholinv", "n": n, "c": c, "bc": -3, "ci": 1, "serialize": False, "policy": 0, "layout": 1},
        {"tag": "qr", "kind": "cacqr", "m": (1 << 13) * world, "n": 256, "serialize": True},
        {"tag": "qr3d", "kind": "cacqr", "m": 1 << 14, "n": 512, "c": 2, "ci": 1, "bc": -1, "serialize": False},
        {"tag": "qr3d_chunks", "kind": "cacqr", "m": 1 << 14, "n": 512, "c": 2, "ci": 0, "bc": -1, "serialize": False, "chunks": 3,
         "env": {"CAPITAL_MULTIPATH_MIN": "4096"}},
    ]
    if mode == "host":          # (the asynchronous run carries every case; the host-staged one, blind to ordering, keeps one of each kind)
        cases = [k for k in cases if k["tag"] in ("ch_p3", "ch_mp_chunks", "ch_l1", "qr3d_chunks")]
    with tempfile.TemporaryDirectory() as d:
        _launch_thread_ranks(4, 2, {"dir": d, "cases": cases}, mode)
        _check_cases(oracle, d, cases, world, c)
