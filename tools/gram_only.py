import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
m, n = 1 << 22, 256
A = torch.rand((n, m), dtype=torch.float64, device="cuda") - 0.5
G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for _ in range(2):
    h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), m, 0.0, capi.ptr(G), n)
h.sync()
