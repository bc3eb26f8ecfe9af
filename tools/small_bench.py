"""Per-call time of the recursion's small products (serialized on the stream), for CAPI_SMALL=0/1 comparisons."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    h.sync()
    ms = C.c_float()
    h.call("capi_timer_start")
    for _ in range(reps): fn()
    h.call("capi_timer_stop_ms", C.byref(ms))
    return ms.value / reps * 1e3
out = []
sizes = [int(x) for x in sys.argv[1:]] or [128, 256, 512, 1024, 2048]
for n in sizes:
    ld = max(4096, n)
    T = torch.rand((ld, ld), dtype=torch.float64, device="cuda")
    B = torch.rand((ld, ld), dtype=torch.float64, device="cuda")
    W = torch.zeros((ld, ld), dtype=torch.float64, device="cuda")
    t1 = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(T), ld, capi.ptr(B), ld, capi.ptr(W), ld))   # L,U,T
    t2 = timeit(lambda: h.call("capi_dgemmt", 1, 1, 0, n, n, -1.0, capi.ptr(B), ld, capi.ptr(B), ld, 1.0, capi.ptr(W), ld))     # U, T N
    t3 = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 0, 0, n, n, 1.0, capi.ptr(T), ld, capi.ptr(B), ld, capi.ptr(W), ld))   # L,U,N
    t4 = timeit(lambda: h.call("capi_dtrmm_oop", 1, 1, 0, 0, n, n, -1.0, capi.ptr(T), ld, capi.ptr(B), ld, capi.ptr(W), ld))  # R,U,N
    t5 = timeit(lambda: h.call("capi_dgemm", 0, 0, n, n, n, 1.0, capi.ptr(T), ld, capi.ptr(B), ld, 0.0, capi.ptr(W), ld))
    print(f"CAPI_SMALL={os.environ.get('CAPI_SMALL','auto')} n={n}: trmmLUT {t1:.1f}  gemmtUTN {t2:.1f}  trmmLUN {t3:.1f}  trmmRUN {t4:.1f}  gemmNN {t5:.1f} us", flush=True)
