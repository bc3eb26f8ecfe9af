"""Blocked diagonal-block routine replayed from a hipGraph (CAPI_GRAPH=1) against plain launches: correctness and time per call."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0, own_stream=True)
for n in (1024, 2048):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    W = S.clone()
    torch.cuda.synchronize()
    ms = C.c_float(); best = 1e9
    for it in range(8):
        W.copy_(S); torch.cuda.synchronize(); h.sync()
        h.call("capi_timer_start"); h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n); h.call("capi_timer_stop_ms", C.byref(ms))
        if it >= 3: best = min(best, ms.value)
    R = torch.triu(W.T)          # column-major upper factor -> row-major
    res = ((R.T @ R) - S).abs().max().item() / S.abs().max().item()
    inv = ((X.T @ R) - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max().item()
    print(f"graph={os.environ.get('CAPI_GRAPH', '0')} n={n}: {best * 1e3:8.1f} us  |R^T R - S| {res:.2e}  |Rinv R - I| {inv:.2e}", flush=True)
