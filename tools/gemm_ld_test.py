"""Does a power-of-two leading dimension hurt the tile kernel (L2 set conflicts)?  Same N = K, lda = N vs N + pad."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
def timeit(fn, reps=3):
    fn(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    return best
for pad in (0, 16, 32, 64, 144):
    ld = n + pad
    A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5     # column-major n x n with leading dimension ld
    Cm = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    B = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
    t1 = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), ld, 1.0, capi.ptr(Cm), ld))
    t2 = timeit(lambda: h.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(A), ld, capi.ptr(B), ld, 0.0, capi.ptr(Cm), ld))
    t3 = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(A), ld, capi.ptr(B), ld, capi.ptr(Cm), ld))
    print(f"n={n} pad={pad}: syrk {n**3/t1/1e9:.1f} TF/s  gemm TN {2*n**3/t2/1e9:.1f} TF/s  trmm LUT {n**3/t3/1e9:.1f} TF/s", flush=True)
    del A, B, Cm
