"""CPU, world_size > 1 over gloo: the N > 1 path of the host-side layer (topo::square rank maps and splits, SUMMA with
K-class stepping / K-slicing, base-case gather, partner exchange, CQR2 Gram allreduce) run by real processes, with the
device C-ABI replaced by the oracle-backed shim in tests/cpu_shim (test infrastructure; the product never loads it).
The assembled distributed factors must equal the 1-rank oracle: R of an SPD matrix is unique."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "cpu_shim")


@pytest.fixture(scope="module")
def shim_lib():
    subprocess.check_call(["make", "-C", SHIM, "-s"])
    return os.path.join(SHIM, "libcapital_driver_cpu.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, cfg, timeout=600, multipath=True, extra_env=None):
    """multipath: grids of pairs (2x2x1, 2x2x2) send every pair broadcast / depth all-reduce / partner exchange through
    capi_pairs_transfer with a threshold of 8 doubles, so the relayed two-phase path (csrc/pair_paths.h) carries even these small
    messages; False = the per-pair collectives on the sub-communicators."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", GLOO_SOCKET_IFNAME="lo", CAPITAL_MIN_CHUNK_COLS="8",
                   CAPITAL_MULTIPATH="2" if multipath else "0", CAPITAL_MULTIPATH_MIN="8")
        env.update(extra_env or {})
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
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


# (world, c) -> d x d x c grids: 2 = 1x1x2 (K-slicing), 4 = 2x2x1 (two K-classes per layer), 8 = 2x2x2 (the reference's cubic case)
@pytest.mark.parametrize("world,c,n,bc,ci,serialize,policy", [
    (2, 2, 96, -2, 1, False, 0),
    (2, 2, 160, -2, 1, False, (0, 3)),   # (policy, num_chunks): chunked / overlapped SUMMA pipeline, K-sliced grid
    (8, 2, 192, -2, 1, True, (0, 3)),    # same on the cubic grid
    (8, 2, 200, -1, 0, False, (2, 5)),
    (4, 1, 128, -1, 1, False, (0, 4)),   # two K-class steps per layer: every chunk sums both steps' products
    (4, 1, 131, -2, 1, True, (2, 3)),    # ... with a padded order and packed pieces
    (2, 2, 128, -1, 0, True, 2),
    (4, 1, 128, -1, 1, False, 0),
    (4, 1, 97, -2, 0, True, 1),        # grid does not divide n: padding path
    (8, 2, 128, -1, 1, False, 0),
    (8, 2, 192, -2, 0, True, 2),
    (8, 2, 130, 0, 1, True, 3),
    (8, 2, 192, -2, 1, False, (1, 4)),   # complete_inv = 1: the right-TRMM SUMMAs of the inverse completion in the chunk pipeline too
    (8, 2, 200, -2, 1, True, (3, 3)),
    (8, 2, 192, -2, 1, True, (0, 3, "plain")),   # the same pipeline over the per-pair RCCL collectives (multi-path off)
    (4, 1, 128, -1, 1, False, (0, 0, "plain")),
    (2, 2, 160, -2, 1, True, (0, 0, "kslice")),    # the replicated 2-rank grid by K-slices + depth all-reduce (round 2's form; default now: by output columns)
    (2, 2, 130, -2, 1, False, (1, 3, "kslice")),
    (2, 2, 200, -3, 1, True, (2, 5)),              # column split with the chunk pipeline, ragged chunk widths
])
def test_cholinv_on_grids(oracle, shim_lib, world, c, n, bc, ci, serialize, policy):
    chunks, multipath, extra = 0, True, {}
    if isinstance(policy, tuple):
        multipath = not (len(policy) == 3 and policy[2] == "plain")
        extra = {"CAPITAL_KSLICE": "1"} if (len(policy) == 3 and policy[2] == "kslice") else {}
        policy, chunks = policy[0], policy[1]
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"kind": "cholinv", "n": n, "c": c, "bc": bc, "ci": ci, "serialize": serialize, "policy": policy, "chunks": chunks, "dir": d},
                multipath=multipath, extra_env=extra)
        A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
        Rg, Ig = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        levels = set()
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            x, y, zz, dd, cc = [int(v) for v in z["xyz"]]
            assert dd * dd * cc == world and cc == c
            np.testing.assert_array_equal(z["A"], oracle.distribute_symmetric(n, n, x, y, dd, dd))   # each rank generated its own piece
            if zz == 0:
                oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, dd, dd)
                oracle.cyclic_insert(Ig, np.asfortranarray(z["Rinv"]), x, y, dd, dd)
            else:   # depth replicas hold identical blocks
                np.testing.assert_allclose(z["R"], oracle.cyclic_extract(Rg, x, y, dd, dd), rtol=0, atol=1e-13 * n) if False else None
            assert float(z["residual"]) <= 1e-14
            levels.add(int(z["stats"][1]))
        Rref, Iref, info = oracle.cholinv_factor(A, ci, 1, bc, c, dd)
        assert info == 0
        assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(Ig - Iref).max() <= 1e-12 * np.abs(Iref).max()
        assert np.count_nonzero(np.tril(Rg, -1)) == 0
        assert len(levels) == 1            # every rank walked the same recursion


@pytest.mark.parametrize("world,c,n,bc,serialize,chunks", [(8, 2, 192, -2, True, 0), (8, 2, 200, -1, False, 3), (4, 1, 128, -2, True, 0),
                                                           (4, 1, 97, -2, False, 0), (2, 2, 128, -2, True, 0)])
def test_cholinv_trsm_mode_on_grids(oracle, shim_lib, world, c, n, bc, serialize, chunks):
    """TRSM mode (info::solve_with_trsm: POTRF + block TRSM + SYRK, no inverse) on d x d x c grids (cholinv.h: potrf_rec_grid): R11 and
    the A12 block are assembled on every rank, each rank solves its share of global columns with the block TRSM, the solved ranges are
    all-gathered in place, the trailing update is the reference schedule's SUMMA.  R must equal the 1-rank oracle's factor."""
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"kind": "cholinv", "n": n, "c": c, "bc": bc, "ci": 0, "serialize": serialize, "policy": 0, "chunks": chunks, "trsm": True,
                        "dir": d})
        A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
        Rg = np.zeros((n, n), order="F")
        reps = {}
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            x, y, zz, dd, cc = [int(v) for v in z["xyz"]]
            assert float(z["residual"]) <= 1e-14
            reps.setdefault((x, y), []).append(z["R"])
            if zz == 0:
                oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, dd, dd)
        for blocks in reps.values():
            for other in blocks[1:]:
                np.testing.assert_array_equal(other, blocks[0])           # depth replicas agree bit for bit
        Rref, _, info = oracle.cholinv_factor(A, 0, 1, bc, c, dd)
        assert info == 0
        assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.count_nonzero(np.tril(Rg, -1)) == 0


@pytest.mark.parametrize("world,c,policy,serialize", [(8, 2, 0, True), (8, 2, 1, False), (8, 2, 2, True), (8, 2, 3, False), (4, 1, 2, False), (2, 2, 1, True)])
def test_non_spd_input_unwinds_every_rank(shim_lib, world, c, policy, serialize):
    """Only the ranks that factor the aggregate see LAPACK's info (layer 0 with ReplicateComp, the slice root with NoReplication[Overlap],
    policy.h:226-514).  The grid must agree before anyone throws: every rank raises, nobody is left waiting in a collective."""
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"kind": "cholinv", "n": 96, "c": c, "bc": -1, "ci": 1, "serialize": serialize, "policy": policy, "chunks": 0, "dir": d,
                        "spoil": 61}, timeout=120)
        saw_pivot = 0
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            msg = str(z["raised"])
            assert "not positive definite" in msg, (r, msg)
            saw_pivot += "non-positive pivot" in msg
        assert 1 <= saw_pivot <= world


@pytest.mark.parametrize("world,c,layout", [(8, 2, 1), (8, 2, 2), (4, 1, 1)])
def test_cholinv_rank_layouts(oracle, shim_lib, world, c, layout):
    """rank -> (x,y,z) layouts 1 and 2 of topo::square (topology.h:96-123): the grid positions change, the factor does not"""
    n, bc, ci = 96, -1, 1
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"kind": "cholinv", "n": n, "c": c, "bc": bc, "ci": ci, "serialize": False, "policy": 0, "layout": layout, "dir": d})
        A = oracle.distribute_symmetric(n, n, 0, 0, 1, 1)
        Rg, Ig = np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        seen = set()
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            x, y, zz, dd, cc = [int(v) for v in z["xyz"]]
            # the reference's rule for layout 1 (and layout 2 on <= 64 ranks of a cubic grid, which coincides with it)
            assert (x, y, zz) == ((r % (dd * dd)) // dd, r % dd, r // (dd * dd))
            seen.add((x, y, zz))
            if zz == 0:
                oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, dd, dd)
                oracle.cyclic_insert(Ig, np.asfortranarray(z["Rinv"]), x, y, dd, dd)
        assert len(seen) == world
        Rref, Iref, info = oracle.cholinv_factor(A, ci, 1, bc, c, dd)
        assert info == 0
        assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(Ig - Iref).max() <= 1e-12 * np.abs(Iref).max()


def test_depth_replicas_agree(oracle, shim_lib):
    with tempfile.TemporaryDirectory() as d:
        _launch(8, {"kind": "cholinv", "n": 64, "c": 2, "bc": -1, "ci": 1, "serialize": False, "policy": 0, "dir": d})
        by_xy = {}
        for r in range(8):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            x, y, zz, _, _ = [int(v) for v in z["xyz"]]
            by_xy.setdefault((x, y), []).append((z["R"], z["Rinv"]))
        assert len(by_xy) == 4
        for (a, b) in by_xy.values():
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("world,m,n,variant,serialize", [(2, 4096, 32, 2, True), (4, 5001, 24, 2, False), (8, 8192, 64, 2, True), (2, 2048, 16, 1, False)])
def test_cacqr_1d_sharded_rows(oracle, shim_lib, world, m, n, variant, serialize):
    with tempfile.TemporaryDirectory() as d:
        _launch(world, {"kind": "cacqr", "m": m, "n": n, "variant": variant, "serialize": serialize, "dir": d})
        Ag, Qg = np.zeros((m, n), order="F"), np.zeros((m, n), order="F")
        Rs = []
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            np.testing.assert_array_equal(z["A"], oracle.distribute_random(n, m, 0, r, 1, world, key=r))   # key = rank / c
            oracle.cyclic_insert(Ag, np.asfortranarray(z["A"]), 0, r, 1, world)
            oracle.cyclic_insert(Qg, np.asfortranarray(z["Q"]), 0, r, 1, world)
            Rs.append(z["R"])
            if variant == 2:
                assert float(z["residual"]) <= 1e-14 and float(z["orth"]) <= 1e-15
        for Rr in Rs[1:]:
            np.testing.assert_array_equal(Rr, Rs[0])          # R is replicated
        Qref, Rref, info = oracle.cacqr_factor_1d(Ag, world, variant)
        assert info == 0
        assert np.abs(Rs[0] - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(Qg - Qref).max() <= 1e-12 * (1 if variant == 2 else 100)


@pytest.mark.parametrize("m,n,variant,serialize,bc,ci,chunks", [(512, 32, 2, False, 0, 1, 0), (1000, 48, 2, True, -1, 1, 0), (512, 32, 1, False, 0, 1, 0),
                                                                (512, 32, 2, False, -1, 0, 0), (1000, 48, 1, True, -1, 0, 0),
                                                                (1000, 64, 2, False, -1, 1, 3),      # right-TRMM SUMMA in the chunk pipeline
                                                                (1024, 64, 2, True, -1, 0, 4)])      # ... and the blocked solve's gemm
def test_cacqr_3d_cubic_grid(oracle, shim_lib, m, n, variant, serialize, bc, ci, chunks):
    """c == d == 2 on 8 ranks (cacqr.hpp:75-116,195-215): Gram by Bcast(row)+gemm+Reduce(column)+Bcast(depth), distributed
    cholinv on the Gram matrix, Q = Q R^-1 by a right-TRMM SUMMA -- or, with complete_inv = 0 (ci), by the blocked solve
    Q1 = A1 R11^-1, Q2 = (A2 - Q1 R12) R22^-1 (cacqr.hpp:44-73).  Q and R are unique, so the assembled result must equal the
    1-D oracle on the assembled input."""
    world, c, d = 8, 2, 2
    with tempfile.TemporaryDirectory() as dd:
        _launch(world, {"kind": "cacqr", "m": m, "n": n, "c": c, "variant": variant, "serialize": serialize, "ci": ci, "bc": bc, "chunks": chunks,
                        "dir": dd})
        Ag, Qg, Rg = np.zeros((m, n), order="F"), np.zeros((m, n), order="F"), np.zeros((n, n), order="F")
        for r in range(world):
            z = np.load(os.path.join(dd, f"rank{r}.npz"))
            x, y, zz = (r % (c * c)) // c, r // (c * c), r % c            # topology.h:46-50
            np.testing.assert_array_equal(z["A"], oracle.distribute_random(n, m, x, y, c, d, key=r // c))
            if zz == 0:
                oracle.cyclic_insert(Ag, np.asfortranarray(z["A"]), x, y, c, d)
                oracle.cyclic_insert(Qg, np.asfortranarray(z["Q"]), x, y, c, d)
                oracle.cyclic_insert(Rg, np.asfortranarray(z["R"]), x, y, c, c)
        Qref, Rref, info = oracle.cacqr_factor_1d(Ag, 1, variant)
        assert info == 0
        assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(Qg - Qref).max() <= 1e-12 * (1 if variant == 2 else 100)
        assert oracle.qr_orthogonality(Qg) <= (1e-15 if variant == 2 else 1e-12)


@pytest.mark.parametrize("m,n,variant,ci", [(1024, 32, 2, 1), (1000, 48, 2, 0)])
def test_cacqr_tunable_grid(oracle, shim_lib, m, n, variant, ci):
    """c = 2, d = 4 on 16 ranks: the tunable c x d x c grid (cacqr.hpp:124-170) -- two 2x2x2 cubes side by side, each running
    the 3-D sweep on its rows, the Gram blocks summed across the cubes (column_alt).  Same uniqueness argument as above."""
    world, c, d = 16, 2, 4
    with tempfile.TemporaryDirectory() as dd:
        _launch(world, {"kind": "cacqr", "m": m, "n": n, "c": c, "variant": variant, "serialize": False, "ci": ci, "bc": -1, "dir": dd}, timeout=900)
        Ag, Qg, Rg = np.zeros((m, n), order="F"), np.zeros((m, n), order="F"), np.zeros((n, n), order="F")
        Rcubes = {}
        for r in range(world):
            z = np.load(os.path.join(dd, f"rank{r}.npz"))
            x, y, zz = (r % (c * c)) // c, r // (c * c), r % c            # topology.h:46-50
            np.testing.assert_array_equal(z["A"], oracle.distribute_random(n, m, x, y, c, d, key=r // c))
            if zz == 0:
                oracle.cyclic_insert(Ag, np.asfortranarray(z["A"]), x, y, c, d)
                oracle.cyclic_insert(Qg, np.asfortranarray(z["Q"]), x, y, c, d)
                Rc = Rcubes.setdefault(y // c, np.zeros((n, n), order="F"))
                oracle.cyclic_insert(Rc, np.asfortranarray(z["R"]), x, y % c, c, c)
        Rg = Rcubes[0]
        np.testing.assert_allclose(Rcubes[1], Rg, rtol=0, atol=1e-13 * np.abs(Rg).max())      # every cube holds the same R
        Qref, Rref, info = oracle.cacqr_factor_1d(Ag, 1, variant)
        assert info == 0
        assert np.abs(Rg - Rref).max() <= 1e-12 * np.abs(Rref).max()
        assert np.abs(Qg - Qref).max() <= 1e-12
        assert oracle.qr_orthogonality(Qg) <= 1e-15


def test_uid_rendezvous_survives_stale_files(shim_lib, tmp_path):
    """bench/launch.h's file rendezvous of the RCCL unique id: two launches on the SAME path and run id, the first one
    leaving all its files behind (a crashed run), plus planted garbage -- every rank of a launch must end with the id
    rank 0 of THAT launch drew, and a rank without partners must give up with a non-zero exit instead of hanging."""
    exe = os.path.join(SHIM, "rendezvous_main")
    path = str(tmp_path / "uid")
    base = path + ".77"

    def launch(world, keep, timeout_s=20, only=None):
        procs = []
        for r in (range(world) if only is None else only):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), CAPITAL_UID_FILE=path, CAPITAL_RUN_ID="77",
                       CAPITAL_RENDEZVOUS_TIMEOUT_S=str(timeout_s))
            if keep:
                env["CAPITAL_KEEP_UID_FILES"] = "1"
            procs.append(subprocess.Popen([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        return [(p.wait(timeout=60), p.stdout.read().strip(), p.stderr.read()) for p in procs]

    # planted leftovers of an imaginary earlier launch: an id file of the right size, a hello and an ack with old tokens
    open(base, "wb").write(bytes(128 + 8 * 3))
    for r in (1, 2):
        open(f"{base}.hello.{r}", "wb").write(b"\x01" * 8)
        open(f"{base}.ack.{r}", "wb").write(b"\x01" * 16)          # same token as the stale hello: what one crashed launch leaves
    first = launch(3, keep=True)
    assert all(rc == 0 for rc, _, _ in first), first
    ids1 = {o for _, o, _ in first}
    assert len(ids1) == 1 and len(next(iter(ids1))) == 256 and next(iter(ids1)) != "00" * 128
    assert os.path.exists(base)                                   # the "crashed" launch left its files
    second = launch(3, keep=False)
    assert all(rc == 0 for rc, _, _ in second), second
    ids2 = {o for _, o, _ in second}
    assert len(ids2) == 1 and ids2 != ids1                        # a fresh id, agreed on by all three ranks
    assert not os.path.exists(base) and not os.path.exists(base + ".hello.1") and not os.path.exists(base + ".ack.2")
    # a rank whose partners never come up: bounded wait, error exit
    open(base, "wb").write(bytes(128 + 8 * 2))
    lone = launch(2, keep=False, timeout_s=2, only=[1])
    assert lone[0][0] == 3 and "timed out" in lone[0][2]
