"""Tall-skinny pieces of CQR2 in isolation: Gram (syrk, K = m) and Q = A * T (right trmm)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pad = int(os.environ.get("TS_PAD", "0"))
ld = m + pad
A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5      # column-major m x n, leading dimension ld
Q = torch.empty((n, ld), dtype=torch.float64, device="cuda")
G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda")).T.contiguous()   # column-major upper
def timeit(fn, reps=5):
    fn(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    return best
tg = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), ld, 0.0, capi.ptr(G), n))
tt = timeit(lambda: h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(T), n, capi.ptr(A), ld, capi.ptr(Q), ld))
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CAPI_") or k.startswith("TS_"))
print(f"[{tag}] gram {tg:.2f} ms ({m*n*n/tg/1e9:.1f} TF/s, {8*m*n/tg/1e6:.0f} GB/s)   trmm-right {tt:.2f} ms ({m*n*n/tt/1e9:.1f} TF/s, {16*m*n/tt/1e6:.0f} GB/s)", flush=True)
