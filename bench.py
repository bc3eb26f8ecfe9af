"""bench.py -- BASELINE.json's metric on MI355X: whole-job TFLOP/s (algorithmic n^3/3) of the recursive Cholesky
with inverse (cholesky::cholinv<...>::factor) at n = 65536, inputs resident in HBM, at 1/2/4/8 GPUs of one node.

  python bench.py --gpus N --steps K --warmup W

N > 1 either arrives launched (torch.distributed.run / mpiexec style: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_* in the environment,
one rank per GPU) or, when WORLD_SIZE is unset, starts its own N ranks the way `mpiexec -n P ./cholinv` starts the reference's bench
(bench/cholesky/cholinv.cpp:8-13): capital_amd/launch.py -- the launching process makes no GPU call, relays rank 0's JSON line,
propagates the first non-zero exit and ends the remaining ranks; it refuses at once when the node has fewer than N devices.

One "step" = one factor() call on the reference's synthetic SPD matrix (distribute_symmetric, structure.hpp:68-103).
The matrix is the metric's own, n = 65536, at EVERY N (strong scaling): N = 1 holds it on one GPU (it fits: 139 GiB with
both factors and the packed copies), N = 8 is BASELINE config 4 (2 x 2 x 2 grid).
The same JSON line also carries
  * `config2`: n = 32768 on one GPU (BASELINE config 2), N = 1 only;
  * `cholesky_trsm_mode`: n = 65536 without forming any inverse (not the reference's schedule; an extra), at every N;
  * `cacqr2`: CA-CholeskyQR2 on m = 2^22 x 256 (BASELINE config 3), N = 1 only;
  * `cacqr2_config5`: CA-CholeskyQR2 with the per-GPU slice of BASELINE config 5, 2^23 x 1024 on every GPU (m = 2^23 N:
    weak in m; N = 8 IS config 5, m = 2^26), 4 m n^2 flops;
  * `roofline`: the dominant kernel (the k-contiguous "TN" 128-tile MFMA kernel that runs the trailing update and the R12
    product): every launch of it inside the timed region stamps its own execution interval (wall clock, first workgroup's start ..
    last workgroup's end); achieved = algorithmic flops / union of those intervals (HIP-event brackets are reported beside it);
  * `cpu_baseline`: the reference's schedules on the node's own host BLAS (oracle.host_baseline, its own interpreter).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Under rocprofv3 a long process must not let a HIP queue's ring wrap: with `--pmc`, librocprofiler-sdk's packet interceptor (ROCm 7.2.0) reads
# past the END of the 1 MiB ring (16384 AQL packets) when the HSA runtime hands it a batch that crosses the wrap -- a host SIGSEGV late in the
# process, located in round 4 (profiles/r4_pmc_whole_bench_crash.txt; rounds 2-3 recorded it as rc 139).  A ring of 131072 packets never wraps
# within a bench process (~50 000 dispatches); set before the HIP runtime creates its queues, and only when a profiler is attached.
if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
    if "ROC_AQL_QUEUE_SIZE" not in os.environ:
        os.environ["ROC_AQL_QUEUE_SIZE"] = "131072"
        sys.stderr.write("bench.py: a rocprofiler tool is attached: ROC_AQL_QUEUE_SIZE=131072 (see profiles/r4_pmc_whole_bench_crash.txt)\n")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak per GPU (vendor figure)

N_CHOLESKY = 65536                # BASELINE.json `metric`: Cholesky n = 65536 -- the same matrix at every N
GRID_C = {1: 1, 2: 2, 4: 1, 8: 2}  # N -> c of the d x d x c grid (d*d*c = N): 1x1x1, 1x1x2, 2x2x1, 2x2x2
QR_CONFIG3 = (1 << 22, 256)       # BASELINE config 3 (one GPU)
QR_CONFIG5_SLICE = (1 << 23, 1024)  # rows per GPU and width of BASELINE config 5 (m = 2^26 on 8 GPUs)
BASE_CASE_ORDER = 1024            # aggregated order of the recursion's base case on the grids (N > 1: 2 x and 4 x this are probed beside it)
BASE_CASE_ORDER_ONE_GPU = 2048    # on one GPU: since round 4's work on the diagonal-block routine an order-2048 base case (1.08 ms) beats two of order 1024 plus the products
                                  # between them; measured on one box each (profiles/r4_base_case_order_one_gpu.txt): n = 65536 1688.9 / 1689.0 -> 1683.3 / 1684.0 ms,
                                  # n = 32768 236.3-237.1 -> 235.4-235.8 ms; order 4096 and 512 lose
T_START, BUDGET_S = 0.0, 480.0    # set in main()


def barrier_sync(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, distributed, device):
    if not distributed:
        return t
    v = torch.tensor([t], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return float(v.item())


def bc_mult_for(n, d, c, order):
    """bc_mult_dim (cholinv.hpp:15-18: t = c d 2^|bc|, bc_loc = n_loc / t, base-case order = d bc_loc) giving `order`."""
    n_loc = -(-n // d)
    b = 0
    while d * (n_loc // (c * d * (1 << (b + 1)))) >= order:
        b += 1
    return -b


def cpu_baseline(n_sample, m_sample, qn):
    """The reference's schedules with their seven BLAS/LAPACK calls bound to the host library of THIS box, all of this
    process's cores, in a separate interpreter (no torch / HIP runtime in it).  Bounded sample of the same workload."""
    cmd = [sys.executable, "-m", "oracle.host_baseline", "--n", str(n_sample), "--m", str(m_sample), "--qn", str(qn)]
    t0 = time.perf_counter()
    try:
        res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        js = json.loads(res.stdout.strip().splitlines()[-1])
    except Exception as e:   # the baseline is a reported extra: its failure must not take the GPU numbers with it
        return {"value": None, "unit": "TFLOP/s", "cores": None, "kind": "port", "sample": f"oracle.host_baseline failed: {e!r}"[:300]}
    ch, qr = js.get("cholesky", {}), js.get("cacqr2", {})
    # `kind` keeps to the contract's two values: "port" = the oracle's restatement of the reference's schedule ("reference" would be the
    # reference's own build, which this image cannot make: no mkl.h).  `blas` says who does the arithmetic inside it.
    return {"value": ch.get("tflops"), "unit": "TFLOP/s", "cores": js["cores"], "kind": "port",
            "blas": "host-blas" if js["kind"] == "host-blas" else "the oracle's own kernels (no host BLAS found)", "library": js["library"],
            "library_path": js["library_path"],
            "sample": f"n={n_sample} recursive Cholesky with inverse (same generator and schedule, bc_mult={ch.get('bc_mult')}), "
                      f"{ch.get('seconds', 0):.2f} s, faster of two runs; whole leg {time.perf_counter() - t0:.0f} s",
            "cacqr2": {"value": qr.get("tflops"), "unit": "TFLOP/s", "sample": f"CholeskyQR2 m={m_sample} n={qn}, {qr.get('seconds', 0):.2f} s"}}


def recorded_traffic(n, gpus):
    """L2-to-fabric bytes per launch of the roofline kernel (both of its symbols) from the committed PMC passes of the same configuration
    (2 x FETCH_SIZE + WRITE_SIZE on gfx950; profiles/README.md).  Counters cannot be read from inside a timed run, so this is
    a RECORDED figure (`traffic_source` names the files), or None when no pass of this configuration AND launch form is committed."""
    import csv
    if (n, gpus) != (65536, 1) or os.environ.get("CAPITAL_NO_LAUNCH_ROUNDS"):
        return None, None          # the committed passes describe the default form: launches in resident rounds, n = 65536 on one GPU
    tag = "r4_pmc_{}_bench_n65536.csv"
    tot, launches = 0.0, 0
    try:
        for t, name in (("fe", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
            rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag.format(t)))) if r["Counter_Name"] == name]
            launches = len(rows)
            tot += sum(float(r["Counter_Value"]) for r in rows) * 1024 * (2 if name == "FETCH_SIZE" else 1)
    except (OSError, KeyError, ValueError):
        return None, None
    if not launches:
        return None, None
    commit = ""
    try:
        commit = open(os.path.join(ROOT, "profiles", "r4_capture_commit.txt")).read().split()[0]
    except (OSError, IndexError):
        pass
    return tot / launches, (f"recorded: profiles/{tag.format('{fe,wr}')}: 2 x FETCH_SIZE + WRITE_SIZE, mean of the {launches} launches (resident rounds) of one factor(), "
                            f"both symbols; {tot / 1e9:.0f} GB per factor()" + (f"; captured at commit {commit}" if commit else ""))


def leg_starts(tag):
    """diagnostics (CAPITAL_BENCH_DUMP_MAPS=<file>): which leg the process is in and, rewritten at every leg, its memory map -- a fault address
    from a profiler's crash report can then be attributed to a module (the rocprofv3 --pmc host SIGSEGV, profiles/README.md)"""
    path = os.environ.get("CAPITAL_BENCH_DUMP_MAPS")
    if not path:
        return
    sys.stderr.write(f"bench.py: leg {tag} starts at +{time.perf_counter() - T_START:.1f} s\n")
    sys.stderr.flush()
    try:
        with open("/proc/self/maps") as f, open(path, "w") as g:
            g.write(f"# /proc/self/maps of pid {os.getpid()} at the start of leg {tag}\n")
            g.write(f.read())
    except OSError:
        pass


def time_cholesky(driver, L, h, n, c, bc, chunks, steps, warmup, distributed, device, bc_policy, trsm_mode=False, multipath=None):
    leg_starts(f"cholesky n={n} grid c={c} chunks={chunks}" + (" TRSM mode" if trsm_mode else ""))
    if multipath is not None:          # read by topo::square when the grid object is built (capital_amd/src/util/topology.h)
        # "kslice": the replicated 2-GPU grid by K-slices + depth all-reduce instead of by output columns (summa.h: colsplit)
        os.environ.pop("CAPITAL_KSLICE", None)
        if multipath == "kslice":
            os.environ["CAPITAL_KSLICE"] = "1"
            multipath = False
        os.environ["CAPITAL_MULTIPATH"] = "1" if multipath else "0"
    prob = driver.Cholinv(n, c=c, complete_inv=0, split=1, bc_mult=bc, layout=0, num_chunks=chunks, serialize=True, bc_policy=bc_policy, trsm_mode=trsm_mode)
    prob.generate()
    for _ in range(warmup):
        prob.factor()
    driver.sync()
    barrier_sync(distributed)
    L.capi_prof_enable(h, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        prob.factor()
    driver.sync()
    barrier_sync(distributed)
    dt = max_over_ranks(time.perf_counter() - t0, distributed, device)
    L.capi_prof_enable(h, 0)
    launches, tot_ms, tot_fl, max_ms = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    L.capi_prof_collect(h, 11, C.byref(launches), C.byref(tot_ms), C.byref(tot_fl), C.byref(max_ms))   # 8 + 3: the 128-tile TN kernel, one symbol
    allv = [C.c_int64(), C.c_double(), C.c_double()]
    L.capi_prof_collect(h, -1, C.byref(allv[0]), C.byref(allv[1]), C.byref(allv[2]), None)
    # the same launches timed by the kernels themselves (first workgroup's start .. last workgroup's end): union and sum of the intervals.
    # 103 = the 128-tile kernels on k-contiguous operands, TWO symbols with one inner loop: dgemm_tile_kernel<128,true,true> (trailing
    # updates; TRMMs when launched one product at a time) and dtrmm_pair_kernel<true,true> (the R12 products as equal-work tile pairs, the form
    # resident rounds need); 11 and 27 = each symbol alone
    def intervals(code):
        v = [C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()]
        L.capi_prof_collect_intervals(h, code, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]), C.byref(v[4]))
        return {"launches": v[0].value, "union_ms": v[1].value, "sum_ms": v[2].value, "flops": v[3].value, "max_ms": v[4].value}
    iv, iv_tile, iv_pair, iva = intervals(103), intervals(11), intervals(27), intervals(-1)
    ms = dt / steps * 1e3
    res = {"ms_per_step": ms, "tflops": n ** 3 / 3.0 / (ms * 1e-3) / 1e12, "residual": prob.residual(), "stats": prob.stats(),
           "grid": [prob.d, prob.d, prob.c],
           "kernel": {"launches": launches.value, "ms": tot_ms.value, "flops": tot_fl.value, "max_ms": max_ms.value, "all_tile_ms": allv[1].value,
                      "iv": iv, "iv_tile": iv_tile, "iv_pair": iv_pair, "iv_all": iva}}
    prob.close()
    return res


def time_cacqr2(driver, m, n, reps, distributed, device):
    leg_starts(f"cacqr2 {m} x {n}")
    q = driver.Cacqr(m, n, c=1, variant=2)
    q.generate()
    q.factor()
    driver.sync()
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for _ in range(reps):
        q.factor()
    driver.sync()
    barrier_sync(distributed)
    dt = max_over_ranks(time.perf_counter() - t0, distributed, device) / reps
    out = {"tflops": 4.0 * m * n * n / dt / 1e12, "ms": dt * 1e3, "residual": q.residual(), "orthogonality": q.orthogonality(), "m_loc": q.m_loc}
    q.close()
    return out


def make_line(args, n, bc, r, chunks, multipath, rccl, residual_max, comm_forms, traffic_rec):
    """the one JSON line (without the extras that main() appends) for a timed Cholesky result `r`"""
    traffic, traffic_src = traffic_rec
    k = r["kernel"]
    # achieved rate of the dominant kernel = its algorithmic flops / the time during which it had workgroups on the device (union of the
    # launches' own execution intervals, capi_prof_collect_intervals).  The launches of the lookahead's bulk streams go out one resident
    # round at a time and interleave: stream-ordered HIP-event brackets then also contain the neighbours' rounds (`by_event_brackets`).
    by_events = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
    iv = k["iv"]
    achieved = iv["flops"] / (iv["union_ms"] * 1e-3) / 1e12 if iv["union_ms"] > 0 else by_events
    steps = max(args.steps, 1)

    def sym(name, v):
        return {"symbol": name, "launches_per_step": v["launches"] / steps, "avg_launch_ms": v["sum_ms"] / max(v["launches"], 1),
                "algorithmic_tflops_per_step": v["flops"] / steps / 1e12,
                "rate_over_own_union": v["flops"] / (v["union_ms"] * 1e-3) / 1e12 if v["union_ms"] > 0 else None}

    out = {
        "metric": "TFLOP/s (whole node) Cholesky n=65536, algorithmic n^3/3, recursive cholinv factor(), inputs resident in HBM",
        "value": r["tflops"], "unit": "TFLOP/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"n={n} recursive Cholesky with inverse (cholinv, complete_inv=0, split=1, bc_mult={bc}) on a "
                               f"{r['grid'][0]}x{r['grid'][1]}x{r['grid'][2]} GPU grid" + (" = BASELINE config 4" if args.gpus == 8 and n == 65536 else ""),
                   "n": n, "grid": r["grid"], "base_case_order": r["stats"]["bc_dimension"], "residual": residual_max,
                   "residual_is": "max over ranks of the reference validator (test/cholesky/validate.hpp:7-49)", "summa_chunks": chunks,
                   "multipath_pair_transfers": multipath is True, "comm_forms": comm_forms,
                   "launches_in_resident_rounds": not os.environ.get("CAPITAL_NO_LAUNCH_ROUNDS"),
                   "rccl_world": list(rccl) if rccl else None},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "timed_by": "the kernels' own wall-clock stamps (first workgroup start .. last workgroup end of every launch); achieved = sum of "
                                 "algorithmic flops / length of the union of the launches' intervals",
                     "kernel": "the 128-tile MFMA kernel on k-contiguous operands (trailing update + R12 product, orders >= 4096): "
                               "dgemm_tile_kernel<128,true,true> and its tile-pair form dtrmm_pair_kernel<true,true>",
                     "symbols": [sym("dgemm_tile_kernel<128, true, true>", k["iv_tile"]), sym("dtrmm_pair_kernel<true, true>", k["iv_pair"])],
                     "launches_per_step": iv["launches"] / steps, "avg_launch_ms": iv["sum_ms"] / max(iv["launches"], 1), "max_launch_ms": iv["max_ms"],
                     "union_ms_per_step": iv["union_ms"] / steps, "sum_ms_per_step": iv["sum_ms"] / steps,
                     "avg_flops_per_launch": iv["flops"] / max(iv["launches"], 1),
                     "by_event_brackets": {"achieved": by_events, "ms_per_step": k["ms"] / steps, "launches": k["launches"],
                                           "note": "HIP events around every launch of dgemm_tile_kernel<128,true,true> alone: stream-ordered, so a bracket "
                                                   "also holds whatever other streams ran between its two events"},
                     "all_tile_kernels_union_share_of_step": k["iv_all"]["union_ms"] / (r["ms_per_step"] * args.steps) if r["ms_per_step"] > 0 else None},
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=0, help="override the Cholesky order (diagnostics only)")
    ap.add_argument("--bc", type=int, default=None, help="bc_mult_dim of cholinv (cholinv.hpp:15-18); default: the value that gives an "
                    "aggregated base-case order of 1024 on this grid (every recursion node below that costs ~13 latency-bound collectives)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-qr", action="store_true")
    ap.add_argument("--no-config2", action="store_true")
    ap.add_argument("--qr-rows", type=int, default=0, help="override the rows per GPU of the config-5 slice (diagnostics / rehearsal only)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: all N ranks on GPU 0 (needs CAPI_RCCL_LIB = the loopback "
                    "library of tests/rccl_loopback; RCCL itself refuses two ranks on one device); never a measurement")
    args = ap.parse_args()
    if args.gpus not in GRID_C:
        raise SystemExit(f"--gpus must be one of {sorted(GRID_C)} (d*d*c grids of one node)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not launched: be the launcher (no GPU call in this process, neither before nor after)
        from capital_amd import launch
        sys.exit(launch.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:], one_device=args.one_device,
                                    timeout_s=float(os.environ.get("CAPITAL_BENCH_WATCHDOG_S", "1500")) + 120))
    from capital_amd import launch
    launch.die_with_parent()
    global T_START, BUDGET_S
    T_START = time.perf_counter()
    # overall budget of one bench process: probes and extras that would start beyond it are skipped and say so in the line
    BUDGET_S = float(os.environ.get("CAPITAL_BENCH_BUDGET_S", "480"))

    # a hung collective must end the run with a traceback and a non-zero exit, not sit on the node until the driver's limit
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("CAPITAL_BENCH_WATCHDOG_S", "1500")), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or unset WORLD_SIZE and let bench.py start them)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product has no CPU path)")
    if args.one_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from capital_amd import capi, driver
    rccl = None
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.one_device:
            dist.init_process_group("gloo")      # torch's side only ships the id and the timing scalars; RCCL proper is the loopback library
        else:
            dist.init_process_group("nccl", device_id=device)
        driver.init_distributed(local_rank)
        rccl = driver.world_query()                 # (rank, size) as RCCL reports them for the world communicator
        assert rccl == (rank, world), f"RCCL world communicator reports rank/size {rccl}, launcher says {(rank, world)}"
    else:
        driver.init(local_rank, 0, 1, None, use_torch_stream=False)
    L = capi.load()
    h = C.c_void_p(driver.handle_ptr())

    n = args.n or N_CHOLESKY
    c = GRID_C[args.gpus]
    d = int(round((args.gpus // c) ** 0.5))
    bc = args.bc if args.bc is not None else bc_mult_for(n, d, c, BASE_CASE_ORDER_ONE_GPU if args.gpus == 1 else BASE_CASE_ORDER)
    # N > 1: how the grid communicates is chosen ON THIS NODE, by measurement.  The plain form -- one RCCL call per
    # pair collective, no overlap -- is timed first, in full (W + K steps): from then on a valid line exists.  Then every other form
    # (multi-path pair transfers over all xGMI links, the chunked SUMMA pipeline on a second stream, both) gets one warm-up and one timed
    # step; the fastest one that validates (residual <= 1e-14) and beats the plain form by > 2 % is timed in full and reported, with the
    # whole table in `config.comm_forms`.  CAPITAL_BENCH_FORM=<name> pins a form (no probing); CAPITAL_BENCH_CHUNKS=<k> pins the pipeline
    # depth of the plain form as before.  Every probe runs under its own timer (CAPITAL_BENCH_PROBE_S, default 120 s, armed and cancelled at
    # the same program points on every rank): if one does not come back, rank 0 prints the plain form's line WITH the hung form recorded
    # (`probe_hang`, config.comm_forms) and every rank exits with code 3.  Probes that would start beyond CAPITAL_BENCH_BUDGET_S are skipped.
    forms = [("plain", int(os.environ.get("CAPITAL_BENCH_CHUNKS", "0")), False)]
    if distributed and "CAPITAL_BENCH_CHUNKS" not in os.environ:
        forms += [("chunks4", 4, False), ("chunks8", 8, False)]
        if args.gpus >= 4:
            forms += [("multipath", 0, True), ("multipath+chunks4", 4, True), ("multipath+chunks8", 8, True)]
        else:
            forms += [("kslice", 0, "kslice"), ("kslice+chunks4", 4, "kslice")]
    pinned = os.environ.get("CAPITAL_BENCH_FORM")
    if pinned:
        forms = [f for f in forms if f[0] == pinned] or forms[:1]
    name, chunks, mp = forms[0]
    r = time_cholesky(driver, L, h, n, c, bc, chunks, args.steps, args.warmup, distributed, device, bc_policy=0 if distributed else 2,
                      multipath=mp if distributed else None)
    comm_forms = [{"form": name, "chunks": chunks, "multipath": mp, "ms_per_step": r["ms_per_step"], "residual": r["residual"], "timed": "full"}]
    if len(forms) > 1:
        import threading
        plain_r = r
        state = {"form": None}

        def bail():
            # A probe did not return: a hung collective of a communication form that has never met this node.  The plain form's measurement is
            # valid and goes out -- WITH the hang recorded in config.comm_forms and in `probe_hang` -- and every rank leaves with exit code 3:
            # the records must show the hang, a clean exit would bury it.
            hung = {"form": state["form"], "error": f"no return within {probe_s:.0f} s (hung collective?)"}
            if rank == 0:
                line = make_line(args, n, bc, plain_r, forms[0][1], forms[0][2], rccl, plain_r["residual"], comm_forms + [hung], recorded_traffic(n, args.gpus))
                line["probe_hang"] = hung
                print(json.dumps(line), flush=True)
            sys.stderr.write(f"bench.py: rank {rank}: probe of form {state['form']!r} did not return within {probe_s:.0f} s; the plain form's line is reported, exit code 3\n")
            sys.stderr.flush()
            os._exit(3)
        probe_s = float(os.environ.get("CAPITAL_BENCH_PROBE_S", "120"))

        def probe(fname, *a, **kw):
            """one probe under its OWN timer, armed on every rank at the same point of the program and cancelled only behind a barrier: either
            every rank's timer is cancelled or every rank's fires (a rank that cancelled early would sit alone in the next collective)"""
            state["form"] = fname
            timer = threading.Timer(probe_s, bail)
            timer.daemon = True
            timer.start()
            try:
                pr = time_cholesky(*a, **kw)
                okv = max_over_ranks(pr["residual"], distributed, device) <= 1e-14
            finally:
                barrier_sync(distributed)
                timer.cancel()
            return pr, okv

        def out_of_budget(what, est_s):
            if time.perf_counter() - T_START + est_s <= BUDGET_S:
                return False
            comm_forms.append({"form": what, "skipped": f"overall budget CAPITAL_BENCH_BUDGET_S={BUDGET_S:.0f} s"})
            return True
        est = 3.0 * r["ms_per_step"] * 1e-3 + 20.0          # a probe = set-up + warm-up + one step (+ validation)
        best = None
        for fname, fch, fmp in forms[1:]:
            if out_of_budget(fname, est):
                continue
            try:
                pr, ok = probe(fname, driver, L, h, n, c, bc, fch, 1, 1, distributed, device, bc_policy=0, multipath=fmp)
                comm_forms.append({"form": fname, "chunks": fch, "multipath": fmp, "ms_per_step": pr["ms_per_step"], "residual": pr["residual"],
                                   "timed": "1 step", "valid": ok})
                if ok and (best is None or pr["ms_per_step"] < best[3]):
                    best = (fname, fch, fmp, pr["ms_per_step"])
            except Exception as e:           # (a driver error raises on every rank alike: the grid agrees on status words)
                comm_forms.append({"form": fname, "chunks": fch, "multipath": fmp, "error": repr(e)[:200]})
        # second stage: the base-case order.  On a grid nothing runs beside the chain of base cases and small SUMMAs (the lookahead is a
        # single-GPU device), and what a collective costs in latency on THIS node decides where the recursion should stop: with the best
        # form so far, aggregated base cases of order 2048 and 4096 (one level, two levels fewer) get a probe each.
        cur = best if (best is not None and best[3] < r["ms_per_step"]) else (name, chunks, mp, r["ms_per_step"])
        best_bc = None
        if args.bc is None and not pinned:
            for order in (2 * BASE_CASE_ORDER, 4 * BASE_CASE_ORDER):
                obc = bc_mult_for(n, d, c, order)
                if obc == bc or out_of_budget(f"{cur[0]} @ base case {order}", est):
                    continue
                try:
                    pr, ok = probe(f"{cur[0]} @ base case {order}", driver, L, h, n, c, obc, cur[1], 1, 1, distributed, device, bc_policy=0, multipath=cur[2])
                    comm_forms.append({"form": cur[0], "chunks": cur[1], "multipath": cur[2], "base_case_order": pr["stats"]["bc_dimension"],
                                       "ms_per_step": pr["ms_per_step"], "residual": pr["residual"], "timed": "1 step", "valid": ok})
                    if ok and pr["ms_per_step"] < cur[3] and (best_bc is None or pr["ms_per_step"] < best_bc[1]):
                        best_bc = (obc, pr["ms_per_step"])
                except Exception as e:
                    comm_forms.append({"form": cur[0], "base_case_order": order, "error": repr(e)[:200]})
        final_ms = best_bc[1] if best_bc else cur[3]
        full_est = (args.steps + args.warmup + 1) * final_ms * 1e-3 + 20.0
        if final_ms < 0.98 * r["ms_per_step"] and not out_of_budget(f"{cur[0]} (full re-run)", full_est):
            name, chunks, mp = cur[:3]
            if best_bc:
                bc = best_bc[0]
            plain = (forms[0][0], forms[0][1], forms[0][2], r, bc_mult_for(n, d, c, BASE_CASE_ORDER) if args.bc is None else args.bc)
            state["form"] = f"{name} (full re-run)"
            timer = threading.Timer(probe_s + full_est, bail)
            timer.daemon = True
            timer.start()
            try:
                r = time_cholesky(driver, L, h, n, c, bc, chunks, args.steps, args.warmup, distributed, device, bc_policy=0, multipath=mp)
            finally:
                barrier_sync(distributed)
                timer.cancel()
            comm_forms.append({"form": name, "chunks": chunks, "multipath": mp, "base_case_order": r["stats"]["bc_dimension"],
                               "ms_per_step": r["ms_per_step"], "residual": r["residual"], "timed": "full"})
            # the full run of the chosen form must validate and must still be the faster one: else the plain form's measurement stands
            if max_over_ranks(r["residual"], distributed, device) > 1e-14 or r["ms_per_step"] >= plain[3]["ms_per_step"]:
                name, chunks, mp, r, bc = plain
                comm_forms.append({"form": name, "note": "reported: the chosen form's full run did not validate or was not faster"})
    residual_max = max_over_ranks(r["residual"], distributed, device)
    out = make_line(args, n, bc, r, chunks, mp, rccl, residual_max, comm_forms if distributed else None, recorded_traffic(n, args.gpus))
    skipped = []

    def in_budget(what, est_s):
        """extras beyond the headline start only while the process is inside CAPITAL_BENCH_BUDGET_S; what is skipped is named in the line
        (every rank takes the same decision at the same point: the clocks differ by less than the margins in the estimates)"""
        t = max_over_ranks(time.perf_counter() - T_START, distributed, device)
        if t + est_s <= BUDGET_S:
            return True
        skipped.append(what)
        return False
    step_s = r["ms_per_step"] * 1e-3

    if args.gpus == 1 and not args.no_config2 and not args.n and in_budget("config2", 12 * step_s / 7 + 20):
        n2 = 32768
        r2 = time_cholesky(driver, L, h, n2, 1, bc_mult_for(n2, 1, 1, BASE_CASE_ORDER_ONE_GPU), 0, max(args.steps, 3), 1, False, device, bc_policy=2)
        k2 = r2["kernel"]
        out["config2"] = {"workload": f"n={n2} recursive Cholesky with inverse on 1 GPU (BASELINE config 2)", "tflops": r2["tflops"],
                          "ms_per_step": r2["ms_per_step"], "residual": r2["residual"],
                          "roofline_kernel_tflops": k2["iv"]["flops"] / (k2["iv"]["union_ms"] * 1e-3) / 1e12 if k2["iv"]["union_ms"] > 0 else None}
        # BASELINE words config 2 as "panel POTRF + TRSM + trailing SYRK": the same matrix in TRSM mode (R only, no inverse formed)
        r2t = time_cholesky(driver, L, h, n2, 1, bc_mult_for(n2, 1, 1, BASE_CASE_ORDER_ONE_GPU), 0, max(args.steps, 3), 1, False, device, bc_policy=2, trsm_mode=True)
        out["config2"]["trsm_mode"] = {"tflops": r2t["tflops"], "ms_per_step": r2t["ms_per_step"], "residual": r2t["residual"]}

    if args.gpus == 1 and not args.no_config2 and not args.n and in_budget("cholesky_trsm_mode", 4 * step_s + 15):
        # NOT the headline: the same matrix factored without forming any inverse (info::solve_with_trsm: potrf + block TRSM + SYRK,
        # what BASELINE north_star names; executes n^3/3 instead of the reference schedule's 5 n^3/12).  Same R, no R^-1.
        rt = time_cholesky(driver, L, h, n, 1, bc, 0, 2, 1, False, device, bc_policy=2, trsm_mode=True)
        out["cholesky_trsm_mode"] = {"workload": f"n={n} Cholesky, TRSM mode (R only, no inverse formed), 1 GPU", "tflops": rt["tflops"],
                                     "ms_per_step": rt["ms_per_step"], "residual": rt["residual"]}

    if distributed and not args.no_config2 and not os.environ.get("CAPITAL_BENCH_NO_TRSM_GRID") and in_budget("cholesky_trsm_mode", 5 * step_s + 20):
        # NOT the headline: TRSM mode on the grid (cholinv.h: potrf_rec_grid; d == 1: every layer factors the replicated matrix) -- an extra;
        # its failure must not take the line with it (a driver error raises on every rank alike)
        try:
            rt = time_cholesky(driver, L, h, n, c, bc, 0, 2, 1, distributed, device, bc_policy=0, trsm_mode=True, multipath=mp)
            out["cholesky_trsm_mode"] = {"workload": f"n={n} Cholesky, TRSM mode (R only, no inverse formed) on the {rt['grid'][0]}x{rt['grid'][1]}x{rt['grid'][2]} grid",
                                         "tflops": rt["tflops"], "ms_per_step": rt["ms_per_step"],
                                         "residual": max_over_ranks(rt["residual"], distributed, device)}
        except Exception as e:
            out["cholesky_trsm_mode"] = {"error": repr(e)[:300]}

    if not args.no_qr and in_budget("cacqr2 / cacqr2_config5", 60):
        reps = max(args.steps, 3)
        if args.gpus == 1:
            m3, n3 = QR_CONFIG3
            q3 = time_cacqr2(driver, m3, n3, reps, False, device)
            out["cacqr2"] = {"workload": f"CA-CholeskyQR2 m={m3} n={n3} (1-D, 1 GPU: BASELINE config 3)", "tflops": q3["tflops"], "ms": q3["ms"],
                             "algorithmic_GBps_per_gpu": 6 * 8.0 * m3 * n3 / (q3["ms"] * 1e-3) / 1e9,
                             "residual": q3["residual"], "orthogonality": q3["orthogonality"]}
        m_loc, n5 = QR_CONFIG5_SLICE
        if args.qr_rows:
            m_loc = args.qr_rows
        elif args.one_device:
            m_loc //= args.gpus            # a rehearsal's ranks share ONE card's HBM: the slice that fills a GPU is divided among them (its rate says nothing)
        m5 = m_loc * args.gpus
        q5 = time_cacqr2(driver, m5, n5, reps, distributed, device)
        out["cacqr2_config5"] = {"workload": f"CA-CholeskyQR2 m={m5} n={n5} (1-D row blocks, {args.gpus} GPU; per-GPU slice {m_loc} x {n5} of BASELINE config 5"
                                             + (": m = 2^26 IS config 5)" if args.gpus == 8 else ")"),
                                 "tflops": q5["tflops"], "ms": q5["ms"], "scaling": "weak (rows per GPU fixed)",
                                 "roofline_frac_of_fp64_mfma": q5["tflops"] / (FP64_MATRIX_PEAK_TFLOPS * args.gpus),
                                 "algorithmic_GBps_per_gpu": 6 * 8.0 * m_loc * n5 / (q5["ms"] * 1e-3) / 1e9,
                                 "residual": max_over_ranks(q5["residual"], distributed, device), "orthogonality": q5["orthogonality"]}

    if skipped:
        out["skipped_for_budget"] = {"budget_s": BUDGET_S, "legs": skipped}
    if rank == 0:
        # register-only MFMA loop, two waves per SIMD, >= 100 ms: what the matrix pipe sustains on THIS device's clocks
        peak = C.c_double()
        L.capi_mfma_f64_peak(h, 250000, C.byref(peak))
        out["mfma_f64_loop_tflops"] = peak.value
    driver.finalize()
    if distributed:
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu and args.gpus == 1:
        out["cpu_baseline"] = cpu_baseline(16384, 1 << 18, 1024)      # ~25 s of host work on the box's CPU share
    if rank == 0:
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
