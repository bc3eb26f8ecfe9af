import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
for (m, n) in ((1 << 20, 256), (1 << 22, 256)):
    A = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5
    tau = torch.zeros(n, dtype=torch.float64, device="cuda")
    ms = C.c_float()
    A0 = A.clone()
    torch.cuda.synchronize()
    for rep in range(2):      # the second pass is timed: the first one grows the handle's workspaces (hipMalloc of tens of GiB)
        A.copy_(A0); torch.cuda.synchronize()
        h.call("capi_timer_start"); h.call("capi_dgeqrf", m, n, capi.ptr(A), m, capi.ptr(tau)); h.call("capi_timer_stop_ms", C.byref(ms)); t1 = ms.value
        h.call("capi_timer_start"); h.call("capi_dorgqr", m, n, n, capi.ptr(A), m, capi.ptr(tau)); h.call("capi_timer_stop_ms", C.byref(ms)); t2 = ms.value
    G = A @ A.T
    print(f"m={m} n={n}: geqrf {t1:.1f} ms ({2*m*n*n/t1/1e9:.2f} TF/s)  orgqr {t2:.1f} ms  |Q^T Q - I| = {(G - torch.eye(n, dtype=torch.float64, device='cuda')).abs().max().item():.2e}", flush=True)
