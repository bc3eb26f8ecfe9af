import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CAPI_LEAF_TRACE"] = "1"
import torch
from capital_amd import capi
h = capi.Handle(0)
n = 128
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for _ in range(3):
    W = S.clone()
    h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n)
h.sync()
