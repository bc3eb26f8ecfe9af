import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for _ in range(3):
    h.call("capi_dgemm", 1, 0, n, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, 0.0, capi.ptr(Cm), n)
h.sync()
