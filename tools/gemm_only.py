"""Trailing-update kernel alone (C(upper) -= A^T A, N = K = 16384: the top level of n = 32768) for PMC collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
for _ in range(3):
    h.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 1.0, capi.ptr(Cm), n)
h.sync()
