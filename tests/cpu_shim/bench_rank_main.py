"""A launched rank for tests/test_launch.py: the bench protocol (barrier, timed factor() calls, max over ranks, ONE JSON line
from rank 0) on the CPU-shim build of the driver over gloo.  Started by capital_amd/launch.py exactly as bench.py's ranks are."""
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    from capital_amd import launch
    launch.die_with_parent()
    mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if mode == "fail" and rank == world - 1:
        print("rank fails on purpose", file=sys.stderr, flush=True)
        sys.exit(7)
    if mode in ("hang", "fail"):
        time.sleep(600)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rank_main
    from capital_amd import driver
    lib = C.CDLL(os.path.join(HERE, "libcapital_driver_cpu.so"), mode=C.RTLD_GLOBAL)
    driver.bind(lib)
    driver._drv = lib
    keep = rank_main.make_callback()
    lib.capi_shim_set_collective(keep)
    assert lib.capital_drv_init(0, rank, world, None, None) == 0
    n, c = 128, {1: 1, 2: 2, 4: 1, 8: 2}[world]
    p = driver.Cholinv(n, c=c, complete_inv=0, split=1, bc_mult=-1, serialize=True, bc_policy=0)
    p.generate()
    p.factor()
    dist.barrier()
    t0 = time.perf_counter()
    p.factor()
    dist.barrier()
    v = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    r = torch.tensor([p.residual()], dtype=torch.float64)
    dist.all_reduce(r, op=dist.ReduceOp.MAX)
    p.close()
    lib.capital_drv_finalize()
    print(f"rank {rank} done", file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps({"metric": "rehearsal", "n_gpus": world, "ms_per_step": float(v) * 1e3, "residual": float(r)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
