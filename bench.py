"""bench.py -- BASELINE.json's metric on MI355X: whole-job TFLOP/s (algorithmic n^3/3) of the recursive Cholesky
with inverse (cholesky::cholinv<...>::factor), inputs resident in HBM, at 1/2/4/8 GPUs of one node.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one factor() call on the reference's synthetic SPD matrix (distribute_symmetric, structure.hpp:68-103).
Per-GPU work is held fixed as N grows (n = 32768 * N^(1/3): weak scaling): N=1 is BASELINE config 2, N=8 config 4.
The same JSON line also carries the CA-CholeskyQR2 figure (config 3 shape per GPU), the roofline of the dominant
kernel (the k-contiguous "TN" MFMA tile kernel that runs the trailing update) measured with HIP events around every
launch in the timed region, and the CPU oracle timed on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak per GPU (vendor figure); cross-checked by capi_mfma_f64_peak

# N -> (n, c): grid is d x d x c with d*d*c = N.  n^3/N is constant across rows (weak scaling in flops).
CHOLESKY_GRID = {1: (32768, 1), 2: (40960, 2), 4: (51200, 1), 8: (65536, 2)}
QR_SHAPE_PER_GPU = (1 << 22, 256)   # BASELINE config 3 on every GPU (m grows with N)


def barrier_sync(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, distributed, device):
    if not distributed:
        return t
    v = torch.tensor([t], dtype=torch.float64, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return float(v.item())


def cpu_baseline(n_sample):
    """The oracle (CPU restatement of the same schedule) on a bounded sample of the same workload."""
    import oracle as O
    O.build()
    threads = min(os.cpu_count() or 1, 16)          # a one-GPU box's CPU share
    O.set_threads(threads)
    A = O.distribute_symmetric(n_sample, n_sample, 0, 0, 1, 1)
    t0 = time.perf_counter()
    R, Ri, info = O.cholinv_factor(A, 0, 1, -3, 1, 1)
    dt = time.perf_counter() - t0
    assert info == 0
    return {"value": (n_sample ** 3 / 3.0) / dt / 1e12, "unit": "TFLOP/s", "cores": threads, "kind": "port",
            "sample": f"n={n_sample} recursive Cholesky, same generator and schedule (bc_mult=-3), oracle/capital_oracle.c, {dt:.1f} s"}


def recorded_traffic(n, gpus):
    """L2-to-fabric bytes per launch of the roofline kernel from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE on
    gfx950; profiles/README.md).  Counters cannot be read from inside a timed run, so this is the last recorded
    measurement of the same configuration, or None.  The passes ran `bench.py --steps 1 --warmup 0`: every row but the
    last (the residual check's product) is a launch of the one factor() call."""
    if n != 32768 or gpus != 1:
        return None, None
    import csv
    tot, launches = 0.0, 0
    try:
        for tag, name in (("fe", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
            rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"r1_d_pmc_{tag}_bench_step.csv"))) if r["Counter_Name"] == name]
            rows = rows[:-1]
            launches = len(rows)
            # KB; gfx950 tallies a 128-B read request at 64 B (double FETCH_SIZE), WRITE_SIZE reads exactly (MI355X_MICROARCH.md)
            tot += sum(float(r["Counter_Value"]) for r in rows) * 1024 * (2 if name == "FETCH_SIZE" else 1)
    except (OSError, KeyError, ValueError):
        return None, None
    if not launches:
        return None, None
    return tot / launches, f"profiles/r1_d_pmc_{{fe,wr}}_bench_step.csv: 2 x FETCH_SIZE + WRITE_SIZE, mean of the {launches} launches of one factor()"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=0, help="override the Cholesky order (diagnostics only)")
    ap.add_argument("--bc", type=int, default=None, help="bc_mult_dim of cholinv (base-case order = c*d*n / 2^|bc| ... see cholinv.hpp:15-18); "
                    "default -5 on one GPU (order 1024), -4 on a grid (aggregated order 1024 as well: every recursion node below "
                    "that costs ~13 latency-bound collectives)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-qr", action="store_true")
    args = ap.parse_args()

    # a hung collective must end the run with a traceback, not sit on the node until the driver's limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("CAPITAL_BENCH_WATCHDOG_S", "1500")), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from capital_amd import capi, driver
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
        driver.init_distributed(local_rank)
    else:
        driver.init(local_rank, 0, 1, None, use_torch_stream=False)
    L = capi.load()
    h = C.c_void_p(driver.handle_ptr())

    if args.gpus not in CHOLESKY_GRID:
        raise SystemExit(f"--gpus must be one of {sorted(CHOLESKY_GRID)} (d*d*c grids of one node)")
    if args.bc is None:
        args.bc = -4 if distributed else -5
    n, c = CHOLESKY_GRID[args.gpus]
    if args.n:
        n = args.n
    # num_chunks > 0 turns on the chunked SUMMA pipeline (collectives on a second HIP stream beside the tile kernel).
    # Off by default until the plain path has been seen to run on a multi-GPU node.
    chunks = int(os.environ.get("CAPITAL_BENCH_CHUNKS", "0")) if distributed else 0
    prob = driver.Cholinv(n, c=c, complete_inv=0, split=1, bc_mult=args.bc, layout=0, num_chunks=chunks, serialize=True, bc_policy=0 if distributed else 2)
    prob.generate()
    for _ in range(args.warmup):
        prob.factor()
    driver.sync()
    barrier_sync(distributed)
    L.capi_prof_enable(h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prob.factor()
    driver.sync()
    barrier_sync(distributed)
    dt = max_over_ranks(time.perf_counter() - t0, distributed, device)
    L.capi_prof_enable(h, 0)
    ms_per_step = dt / args.steps * 1e3
    flops = n ** 3 / 3.0
    value = flops / (ms_per_step * 1e-3) / 1e12

    # roofline of the dominant kernel: every launch of the 128-tile TN MFMA kernel in the timed region, HIP events on its stream
    launches, tot_ms, tot_fl, max_ms = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    L.capi_prof_collect(h, 11, C.byref(launches), C.byref(tot_ms), C.byref(tot_fl), C.byref(max_ms))   # 8 + 3: the 128-tile TN kernel, one symbol
    achieved = tot_fl.value / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else 0.0
    allv = [C.c_int64(), C.c_double(), C.c_double()]
    L.capi_prof_collect(h, -1, C.byref(allv[0]), C.byref(allv[1]), C.byref(allv[2]), None)
    residual = prob.residual()
    stats = prob.stats()
    traffic, traffic_src = recorded_traffic(n, args.gpus)
    prob.close()

    out = {
        "metric": "TFLOP/s (whole node) Cholesky n^3/3, recursive cholinv factor(), inputs resident in HBM",
        "value": value, "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"n={n} recursive Cholesky with inverse (cholinv, complete_inv=0, split=1, bc_mult={args.bc}) on a {prob.d}x{prob.d}x{prob.c} GPU grid",
                   "n": n, "grid": [prob.d, prob.d, prob.c], "base_case_order": stats["bc_dimension"], "residual": residual, "summa_chunks": chunks},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "dgemm_tile_kernel<128,true,true> (trailing update + R12 solve, orders >= 4096)",
                     "launches_per_step": launches.value / max(args.steps, 1),
                     "avg_launch_ms": tot_ms.value / max(launches.value, 1), "max_launch_ms": max_ms.value,
                     "avg_flops_per_launch": tot_fl.value / max(launches.value, 1),
                     "tile_kernel_share_of_step": allv[1].value / (ms_per_step * args.steps) if ms_per_step > 0 else None},
    }

    if not args.no_qr:
        m_loc, nq = QR_SHAPE_PER_GPU
        m = m_loc * args.gpus
        q = driver.Cacqr(m, nq, c=1, variant=2)
        q.generate()
        q.factor()
        driver.sync()
        barrier_sync(distributed)
        t0 = time.perf_counter()
        reps = max(args.steps, 3)
        for _ in range(reps):
            q.factor()
        driver.sync()
        barrier_sync(distributed)
        dtq = max_over_ranks(time.perf_counter() - t0, distributed, device) / reps
        out["cacqr2"] = {"workload": f"CA-CholeskyQR2 m={m} n={nq} (1-D, {args.gpus} GPU)", "tflops": 4.0 * m * nq * nq / dtq / 1e12,
                         "ms": dtq * 1e3, "algorithmic_GBps_per_gpu": 6 * 8.0 * m_loc * nq / dtq / 1e9,
                         "residual": q.residual(), "orthogonality": q.orthogonality()}
        q.close()

    if rank == 0 and not args.no_cpu and args.gpus == 1:
        out["cpu_baseline"] = cpu_baseline(10240)      # ~10 s of host work on 16 threads
    if rank == 0:
        peak = C.c_double()
        L.capi_mfma_f64_peak(h, 20000, C.byref(peak))
        out["mfma_f64_loop_tflops"] = peak.value
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()
    driver.finalize()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
