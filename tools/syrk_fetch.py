"""The trailing updates of the headline step, alone: C(upper) -= A^T A (capi_dgemmt TN) at N x K = 16384 x 16384, 16384 x 32768, 32768 x 32768
and the lookahead's rectangle (capi_dgemm TN, 16384 x 16384 x 32768): time per launch; under rocprofv3 --pmc FETCH_SIZE what they fetch.
CAPI_ROUNDS: bit 0 plain products, bit 1 triangular outputs one resident round per launch.   python tools/syrk_fetch.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("CAPI_ROUNDS",) if k in os.environ) or "default"
def timeit(f):
    f(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(3):
        h.call("capi_timer_start"); f(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    return best
for (N, K) in ((16384, 16384), (16384, 32768), (32768, 32768)):
    A = torch.rand((N, K), dtype=torch.float64, device="cuda")          # column-major K x N
    Cm = torch.zeros((N, N), dtype=torch.float64, device="cuda")
    t = timeit(lambda: h.call("capi_dgemmt", 1, 1, 0, N, K, -1.0, capi.ptr(A), K, capi.ptr(A), K, 1.0, capi.ptr(Cm), N))
    print(f"[{tag}] gemmt upper TN N={N} K={K}: {t:.2f} ms  {float(N) * (N + 1) * K / t / 1e9:.2f} TF/s", flush=True)
    del A, Cm; torch.cuda.empty_cache()
M, N, K = 16384, 16384, 32768
A = torch.rand((M, K), dtype=torch.float64, device="cuda"); B = torch.rand((N, K), dtype=torch.float64, device="cuda")
Cm = torch.zeros((N, M), dtype=torch.float64, device="cuda")
t = timeit(lambda: h.call("capi_dgemm", 1, 0, M, N, K, -1.0, capi.ptr(A), K, capi.ptr(B), K, 1.0, capi.ptr(Cm), M))
print(f"[{tag}] gemm TN {M} x {N} x {K}: {t:.2f} ms  {2.0 * M * N * K / t / 1e9:.2f} TF/s", flush=True)
