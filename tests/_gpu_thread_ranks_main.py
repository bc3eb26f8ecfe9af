"""Ranks as THREADS (tests/test_gpu_multirank.py::test_eight_ranks_as_threads_*): process p of NPROC hosts the ranks p T .. p T + T - 1 of a
world of NPROC * T, one thread each, all on GPU 0, through the thread-ranks build of the host layer (tests/thread_ranks: per-rank state
thread-local) and the loopback transport.  Eight ranks then need four GPU processes (two threads each): the 2 x 2 x 2 grid, the 3-D CholeskyQR2 and the multi-path
transfer sets at P = 8 run on a real MI355X inside the pool's limit of six GPU processes.  No torch.distributed: the unique id travels through
a file, the collectives themselves order the ranks."""
import ctypes as C
import json
import os
import sys
import threading
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


T0 = time.time()


def main():
    cfg = json.loads(sys.argv[1])
    proc, nproc, T = int(os.environ["PROC_INDEX"]), int(os.environ["NPROC"]), int(os.environ["THREADS_PER_PROC"])
    world = nproc * T
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from capital_amd import capi, driver
    L = capi.load()
    D = driver.load()
    assert L.capi_comm_load_rccl(os.environ["CAPI_RCCL_LIB"].encode()) == 0        # bound once, before the threads exist
    uid_path = os.path.join(cfg["dir"], "unique_id.bin")
    if proc == 0:
        buf = (C.c_char * 128)()
        assert L.capi_comm_unique_id(buf) == 0
        with open(uid_path + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.rename(uid_path + ".tmp", uid_path)
    t_end = time.time() + 120
    while not os.path.exists(uid_path):
        assert time.time() < t_end, "the unique id never arrived"
        time.sleep(0.01)
    uid = open(uid_path, "rb").read()
    bar = threading.Barrier(T)
    failures = []

    def rank_main(t):
        rank = proc * T + t
        try:
            ubuf = (C.c_char * 128).from_buffer_copy(uid)
            driver._ck(D.capital_drv_init(0, rank, world, ubuf, None), "capital_drv_init")
            r, s = C.c_int(), C.c_int()
            driver._ck(D.capital_drv_world_query(C.byref(r), C.byref(s)), "world_query")
            assert (r.value, s.value) == (rank, world), f"the transport reports {(r.value, s.value)}, the launcher {(rank, world)}"
            for case in cfg["cases"]:
                tag = case["tag"]
                bar.wait()                               # every thread of the process is between two cases: the environment may change
                if t == 0:
                    for k in ("CAPITAL_MULTIPATH", "CAPITAL_MULTIPATH_MIN"):
                        os.environ.pop(k, None)
                    os.environ.update(case.get("env", {}))
                    print(f"process {proc}: case {tag} starts at +{time.time() - T0:.1f} s", flush=True)
                bar.wait()
                if case["kind"] == "cholinv":
                    p = driver.Cholinv(case["n"], c=case["c"], complete_inv=case["ci"], split=1, bc_mult=case["bc"], layout=case.get("layout", 0),
                                       num_chunks=case.get("chunks", 0), serialize=case["serialize"], bc_policy=case["policy"], trsm_mode=case.get("trsm", False))
                    p.generate()
                    t0 = time.time()
                    p.factor()
                    t1 = time.time()
                    p.factor()
                    res = p.residual()
                    if case.get("light"):            # a large rehearsal (tools/rehearse_config4_grid.py): no factors on disk, their sums and the validator's residual
                        R = p.R()
                        np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), xyz=np.array([p.x, p.y, p.z, p.d, p.c]), residual=res,
                                 sums=np.array([R.sum(), np.abs(R).sum(), float(np.square(R).sum())]), seconds=np.array([t1 - t0, time.time() - t1]))
                        p.close()
                        continue
                    Rinv = p.Rinv() if not case.get("trsm", False) else np.zeros((1, 1))
                    np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), R=p.R(), Rinv=Rinv, xyz=np.array([p.x, p.y, p.z, p.d, p.c]), residual=res)
                    p.close()
                else:
                    q = driver.Cacqr(case["m"], case["n"], c=case.get("c", 1), variant=2, complete_inv=case.get("ci", 0), bc_mult=case.get("bc", 0),
                                     num_chunks=case.get("chunks", 0), serialize=case["serialize"])
                    q.generate()
                    q.factor()
                    c3 = case.get("c", 1)
                    if case.get("light"):
                        R = q.R()
                        np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), residual=q.residual(), orth=q.orthogonality(),
                                 sums=np.array([R.sum(), np.abs(R).sum(), float(np.square(R).sum())]))
                        q.close()
                        continue
                    np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), A=q.A(), Q=q.Q(), R=q.R(),
                             residual=q.residual() if c3 == 1 else -1.0, orth=q.orthogonality() if c3 == 1 else -1.0)
                    q.close()
            bar.wait()
            driver._ck(D.capital_drv_finalize(), "finalize")
            print(f"rank {rank} ok at +{time.time() - T0:.1f} s", flush=True)
        except BaseException as e:          # a rank that fails must take the process down: its peers would wait in a collective
            failures.append((rank, repr(e)))
            traceback.print_exc()
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(17)

    threads = [threading.Thread(target=rank_main, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
