"""cpu_baseline leg of bench.py: the reference's schedules (cholinv.hpp:6-183, cacqr.hpp:7-29,174-193, restated in
capital_oracle.c) with their seven BLAS/LAPACK calls bound to the HOST library found on this machine (libmkl_rt, OpenBLAS,
or numpy/scipy's bundled OpenBLAS -- SURVEY.md 8(d)(i)), on all of this process's cores.  Runs in its own interpreter
(no torch, no GPU, one OpenMP runtime) and prints one JSON object.

    python -m oracle.host_baseline --n 16384 --m 262144 --qn 1024 [--threads T]

TEST / MEASUREMENT INFRASTRUCTURE: this is the checker's side, never the product."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cpu_share():
    """Cores this process may actually use: the cgroup's CPU quota when there is one, else the affinity mask -- capped at 16, a
    one-GPU box's share of its host (the mask there lists all 128 hardware threads; 128 MKL threads on a 16-core share ran
    the n = 16384 schedule at 0.09 TFLOP/s).  CAPITAL_CPU_THREADS overrides."""
    if os.environ.get("CAPITAL_CPU_THREADS"):
        return int(os.environ["CAPITAL_CPU_THREADS"])
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--bc", type=int, default=-4)
    ap.add_argument("--m", type=int, default=1 << 18)
    ap.add_argument("--qn", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-host-blas", action="store_true")
    a = ap.parse_args()
    import oracle as O
    O.build()
    threads = a.threads or cpu_share()
    O.set_threads(threads)
    lib = None if a.no_host_blas else O.bind_host_blas(threads)
    out = {"kind": "host-blas" if lib else "port", "cores": threads if not lib else lib["threads"],
           "library": lib["library"] if lib else "oracle/capital_oracle.c own kernels", "library_path": lib["path"] if lib else None}
    # Every schedule runs twice and the faster run is reported: the first pass pays for first-touch page faults of its
    # buffers and the library's own set-up (seconds in a sandboxed container), which is not what the baseline is about.
    if a.n > 0:
        A = O.distribute_symmetric(a.n, a.n, 0, 0, 1, 1)
        best = None
        import numpy as np
        R, Ri = np.zeros((a.n, a.n), order="F"), np.zeros((a.n, a.n), order="F")
        for _ in range(2):
            t0 = time.perf_counter()
            R, Ri, info = O.cholinv_factor(A, 0, 1, a.bc, 1, 1, out=(R, Ri))
            dt = time.perf_counter() - t0
            assert info == 0
            best = dt if best is None else min(best, dt)
        out["cholesky"] = {"n": a.n, "seconds": best, "tflops": a.n ** 3 / 3.0 / best / 1e12, "bc_mult": a.bc,
                           "residual": O.cholesky_residual(A, R) if a.n <= 4096 else None}
        del A, R, Ri
    if a.m > 0:
        Q = O.distribute_random(a.qn, a.m, 0, 0, 1, 1, key=0)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            Qo, Ro, info = O.cacqr_factor_1d(Q, 1, 2)
            dt = time.perf_counter() - t0
            assert info == 0
            best = dt if best is None else min(best, dt)
        out["cacqr2"] = {"m": a.m, "n": a.qn, "seconds": best, "tflops": 4.0 * a.m * a.qn ** 2 / best / 1e12}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
