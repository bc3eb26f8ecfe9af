"""capi_dpotrf_trtri(n) timing (CAPI_POTRF_TRTRI=rec|blocked selects the schedule)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
out = []
for n in (512, 1024, 2048, 4096, 8192):
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    W = S.clone()
    best = 1e9
    ms = C.c_float()
    for _ in range(4):
        W.copy_(S); h.sync()
        h.call("capi_timer_start"); h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n); h.call("capi_timer_stop_ms", C.byref(ms))
        best = min(best, ms.value)
    R = torch.triu(W.T)                                  # column-major upper -> logical
    err = ((R.T @ R) - S).abs().max().item() / n
    inv = (R @ torch.triu(X.T) - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max().item()
    out.append(f"n={n}: {best*1e3:.0f} us (|R^T R - A|/n {err:.1e}, |R X - I| {inv:.1e})")
print(f"[{os.environ.get('CAPI_POTRF_TRTRI', 'auto')}] " + "  ".join(out), flush=True)
