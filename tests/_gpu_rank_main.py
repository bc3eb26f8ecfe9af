"""One rank of an N-GPU parity run (tests/test_gpu_multirank.py): a fresh process per GPU, torch.distributed (gloo) only to
ship the RCCL unique id, everything else through the product's driver -- RCCL communicators, topo::square / topo::rect splits,
SUMMA collectives, base-case gathers, the CQR2 Gram all-reduce.  Saves this rank's blocks for the parent to assemble."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = json.loads(sys.argv[1])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    # a rank that dies must not leave its peers in gloo's 30-minute default wait
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get("CAPITAL_TEST_GLOO_TIMEOUT_S", "300"))))
    from capital_amd import driver
    driver.init_distributed(local)
    rr, rs = driver.world_query() if world > 1 or os.environ.get("CAPI_RCCL_FORCE") else (rank, world)
    assert (rr, rs) == (rank, world), f"RCCL reports rank/size {(rr, rs)}, launcher {(rank, world)}"
    for case in cfg["cases"]:
        tag = case["tag"]
        for k in ("CAPITAL_MULTIPATH", "CAPITAL_MULTIPATH_MIN"):       # per-case switches, read when the grid object is built
            os.environ.pop(k, None)
        os.environ.update(case.get("env", {}))
        print(f"rank {rank}: case {tag} starts", flush=True)
        if case["kind"] == "cholinv":
            p = driver.Cholinv(case["n"], c=case["c"], complete_inv=case["ci"], split=1, bc_mult=case["bc"], layout=case.get("layout", 0),
                               num_chunks=case.get("chunks", 0), serialize=case["serialize"], bc_policy=case["policy"], trsm_mode=case.get("trsm", False))
            p.generate()
            p.factor()
            p.factor()                      # a second call reuses communicators, streams, events and workspaces
            res = p.residual()
            Rinv = p.Rinv() if not case.get("trsm", False) else np.zeros((1, 1))     # TRSM mode forms no inverse
            np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), R=p.R(), Rinv=Rinv, xyz=np.array([p.x, p.y, p.z, p.d, p.c]),
                     residual=res, stats=np.array(list(p.stats().values())))
            p.close()
        else:
            q = driver.Cacqr(case["m"], case["n"], c=case.get("c", 1), variant=2, complete_inv=case.get("ci", 0), bc_mult=case.get("bc", 0),
                             num_chunks=case.get("chunks", 0), serialize=case["serialize"])
            q.generate()
            q.factor()
            c3 = case.get("c", 1)
            np.savez(os.path.join(cfg["dir"], f"{tag}_rank{rank}.npz"), A=q.A(), Q=q.Q(), R=q.R(),
                     residual=q.residual() if c3 == 1 else -1.0, orth=q.orthogonality() if c3 == 1 else -1.0)
            q.close()
        dist.barrier()
    driver.finalize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
