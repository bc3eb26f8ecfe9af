import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
W = S.clone()
for _ in range(3):
    W.copy_(S)
    h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n)
h.sync()
