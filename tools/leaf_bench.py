import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
def timeit(fn, reps=20):
    fn(); h.sync()
    ms = C.c_float()
    h.call("capi_timer_start")
    for _ in range(reps): fn()
    h.call("capi_timer_stop_ms", C.byref(ms))
    return ms.value / reps * 1000
for n in (16, 32, 64, 128):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    W = S.clone()
    t_full = timeit(lambda: h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n))
    W.copy_(S)
    t_potrf = timeit(lambda: h.call("capi_dpotrf", 1, n, capi.ptr(W), n))
    t_trtri = timeit(lambda: h.call("capi_dtrtri", 1, 0, n, capi.ptr(X), n))
    print(f"n={n}: potrf+trtri {t_full:.1f} us  potrf {t_potrf:.1f} us  trtri {t_trtri:.1f} us", flush=True)
