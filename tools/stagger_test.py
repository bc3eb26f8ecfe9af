import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = 8192
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
def timeit(fn, reps=4):
    fn(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    return best
n2 = 16384
W = torch.rand((n, n2), dtype=torch.float64, device="cuda") - 0.5
S = torch.zeros((n2, n2), dtype=torch.float64, device="cuda")
for rnd in range(2):
    for st in (0,):
        os.environ["CAPI_STAGGER"] = str(st)
        t = timeit(lambda: h.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, 0.0, capi.ptr(Cm), n))
        t2 = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, capi.ptr(Cm), n))
        t3 = timeit(lambda: h.call("capi_dgemmt", 1, 1, 0, n2, n, -1.0, capi.ptr(W), n, capi.ptr(W), n, 1.0, capi.ptr(S), n2), reps=3)
        print(f"prio mode={st}: gemm TN 8192 {2*n**3/t/1e9:.2f} TF/s   trmm LUT {n**3/t2/1e9:.2f} TF/s   gemmt 16384x8192 {n2*(n2+1)*n/t3/1e9:.2f} TF/s", flush=True)
