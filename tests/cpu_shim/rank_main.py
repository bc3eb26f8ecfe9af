"""One rank of a multi-process CPU rehearsal of the host-side layer (tests/test_multirank_gloo.py).

The driver (capital_amd/drivers/capital_driver.cpp + capital_amd/src) is linked against the oracle-backed CPU shim
(tests/cpu_shim/capi_cpu_shim.cpp); its collectives arrive here as callbacks and are carried by torch.distributed
(gloo) point-to-point messages between the world ranks a communicator lists."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def view(ptr, count):
    return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,)))


def make_callback():
    CB = C.CFUNCTYPE(C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

    def cb(op, ranks_p, n, me, buf, buf2, count, root):
        try:
            ranks = [ranks_p[i] for i in range(n)]
            if op == 0:      # bcast
                t = view(buf, count)
                if me == root:
                    for i, r in enumerate(ranks):
                        if i != me:
                            dist.send(t, r)
                else:
                    dist.recv(t, ranks[root])
            elif op == 1:    # allreduce(sum), summed in rank order on the first member, then broadcast
                t = view(buf, count)
                if me == 0:
                    tmp = torch.empty_like(t)
                    for r in ranks[1:]:
                        dist.recv(tmp, r)
                        t += tmp
                    for r in ranks[1:]:
                        dist.send(t, r)
                else:
                    dist.send(t, ranks[0])
                    dist.recv(t, ranks[0])
            elif op == 2:    # reduce send -> recv on root
                s = view(buf, count)
                if me == root:
                    out = view(buf2, count)
                    acc = s.clone()
                    tmp = torch.empty_like(s)
                    for i, r in enumerate(ranks):
                        if i != me:
                            dist.recv(tmp, r)
                            acc += tmp
                    out.copy_(acc)
                else:
                    dist.send(s, ranks[root])
            elif op == 3:    # allgather: count doubles per member
                s = view(buf, count).clone()
                out = view(buf2, count * n)
                if me == 0:
                    out[0:count] = s
                    tmp = torch.empty_like(s)
                    for i, r in enumerate(ranks[1:], start=1):
                        dist.recv(tmp, r)
                        out[i * count:(i + 1) * count] = tmp
                    for r in ranks[1:]:
                        dist.send(out, r)
                else:
                    dist.send(s, ranks[0])
                    dist.recv(out, ranks[0])
            elif op == 4:    # sendrecv_replace with member `root`
                t = view(buf, count)
                peer = ranks[root]
                snd = t.clone()
                if ranks[me] < peer:
                    dist.send(snd, peer)
                    dist.recv(t, peer)
                else:
                    dist.recv(t, peer)
                    dist.send(snd, peer)
            elif op == 5:    # one group of point-to-point messages (capi_pairs_transfer): post everything, then wait
                class Entry(C.Structure):
                    _fields_ = [("ptr", C.c_void_p), ("count", C.c_int64), ("peer", C.c_int32), ("is_send", C.c_int32)]
                ents = C.cast(buf, C.POINTER(Entry))
                reqs, keep_alive = [], []
                for i in range(count):
                    e = ents[i]
                    t = view(e.ptr, e.count)
                    if e.is_send:
                        t = t.clone()           # (the group's sends read their buffers at posting time, as a stream-ordered send does)
                        keep_alive.append(t)
                        reqs.append(dist.isend(t, e.peer))
                    else:
                        reqs.append(dist.irecv(t, e.peer))
                for r in reqs:
                    r.wait()
            else:
                return 1
            return 0
        except Exception as e:  # pragma: no cover
            print("collective callback failed:", repr(e), flush=True)
            return 1

    return CB(cb)


def main():
    cfg = json.loads(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from capital_amd import driver
    lib = C.CDLL(os.environ.get("CAPITAL_SHIM_LIB", os.path.join(HERE, "libcapital_driver_cpu.so")), mode=C.RTLD_GLOBAL)   # (override: the sanitizer build)
    driver.bind(lib)
    driver._drv = lib                      # the driver classes now talk to the CPU-shim build
    keep = make_callback()
    lib.capi_shim_set_collective(keep)
    assert lib.capital_drv_init(0, rank, world, None, None) == 0, lib.capital_drv_last_error()
    out = {"rank": rank}
    if cfg["kind"] == "cholinv":
        p = driver.Cholinv(cfg["n"], c=cfg["c"], complete_inv=cfg["ci"], split=cfg.get("split", 1), bc_mult=cfg["bc"],
                           layout=cfg.get("layout", 0), num_chunks=cfg.get("chunks", 0), serialize=cfg["serialize"], bc_policy=cfg["policy"],
                           trsm_mode=cfg.get("trsm", False))
        p.generate()
        if "spoil" in cfg:
            # a non-SPD input: global diagonal element g made negative on the rank that owns it (local (i,j) <-> global (x + i d, y + j d))
            g = cfg["spoil"]
            if p.x == g % p.d and p.y == g % p.d:
                Aloc = p.A()
                Aloc[g // p.d, g // p.d] = -1.0
                p.set_A(Aloc)
            raised = ""
            try:
                p.factor()
            except driver.DriverError as e:
                raised = str(e)
            # every rank must come back (no one left in a collective) and every rank must have been told
            dist.barrier()
            np.savez(os.path.join(cfg["dir"], f"rank{rank}.npz"), raised=np.array(raised), xyz=np.array([p.x, p.y, p.z, p.d, p.c]))
            p.close()
            lib.capital_drv_finalize()
            dist.barrier()
            dist.destroy_process_group()
            return
        p.factor()
        res = p.residual()
        Rinv = p.Rinv() if not cfg.get("trsm", False) else np.zeros((1, 1))      # TRSM mode forms no inverse
        np.savez(os.path.join(cfg["dir"], f"rank{rank}.npz"), A=p.A(), R=p.R(), Rinv=Rinv, xyz=np.array([p.x, p.y, p.z, p.d, p.c]),
                 residual=res, stats=np.array(list(p.stats().values())))
        p.close()
    else:
        q = driver.Cacqr(cfg["m"], cfg["n"], c=cfg.get("c", 1), variant=cfg["variant"], complete_inv=cfg.get("ci", 0), bc_mult=cfg.get("bc", 0),
                         num_chunks=cfg.get("chunks", 0), serialize=cfg["serialize"])
        q.generate()
        q.factor()
        c3 = cfg.get("c", 1)
        np.savez(os.path.join(cfg["dir"], f"rank{rank}.npz"), A=q.A(), Q=q.Q(), R=q.R(),
                 residual=q.residual() if c3 == 1 else -1.0, orth=q.orthogonality() if c3 == 1 else -1.0)
        q.close()
    lib.capital_drv_finalize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
