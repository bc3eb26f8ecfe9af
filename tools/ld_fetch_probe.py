"""FETCH_SIZE of the tile kernel against the leading dimension (run under rocprofv3 --pmc FETCH_SIZE): one dgemm (TN), one dsyrk,
one dtrmm of order n with all three matrices at leading dimension ld."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]); ld = int(sys.argv[2])
A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
h.call("capi_dgemm", 1, 0, n, n, n, -1.0, capi.ptr(A), ld, capi.ptr(B), ld, 1.0, capi.ptr(Cm), ld)
h.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), ld, 1.0, capi.ptr(Cm), ld)
h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(A), ld, capi.ptr(B), ld, capi.ptr(Cm), ld)
h.sync()
