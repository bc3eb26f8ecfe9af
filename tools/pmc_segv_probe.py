"""Round-2 record gpurun_out/ab4/bench_pmc1.err: a HOST SIGSEGV under `rocprofv3 --pmc` inside the launch of
dgemm_small_kernel<false,true> reached through factor_trsm -> capi_dpotrf -> potrf_trtri_rec -> capi_dtrmm_oop (the first TRSM-mode leg of
bench.py, after ~50 k dispatches of the legs before it).  This probe runs ONLY that leg, first thing in a fresh process, on the call path
the record shows (CAPI_POTRF_DIAG=rec restores it), with every small-kernel launch's arguments on stderr (CAPI_DEBUG_GEMM):
  * crashes again  -> the launch itself (its arguments are the last line of the log);
  * passes         -> not this launch: cumulative state of the profiler over a long run.
usage: [rocprofv3 --pmc FETCH_SIZE -d DIR --] python tools/pmc_segv_probe.py [n] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from capital_amd import driver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
driver.init(0, 0, 1, None, use_torch_stream=False)
p = driver.Cholinv(n, c=1, complete_inv=0, split=1, bc_mult=-5, serialize=True, bc_policy=2, trsm_mode=True)
p.generate()
for i in range(steps):
    t0 = time.perf_counter()
    p.factor()
    driver.sync()
    print(f"probe: TRSM-mode factor n={n} step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
print(f"probe: residual {p.residual():.3e}", flush=True)
p.close()
driver.finalize()
print("probe ok", flush=True)
