"""CPU time to ENQUEUE capi_dpotrf_trtri(n) (asynchronous launches) vs GPU time to execute it: is the chain launch-bound on the host?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
for n in (1024, 2048, 4096):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    W = S.clone()
    for it in range(3):
        W.copy_(S); h.sync()
        t0 = time.perf_counter()
        h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n)
        t1 = time.perf_counter()
        h.sync()
        t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6*(t1-t0):.0f} us, until done {1e6*(t2-t0):.0f} us", flush=True)
