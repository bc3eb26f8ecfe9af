"""Correctness of the tall-skinny kernels against torch (fp64) on a ragged tall shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
torch.manual_seed(1)
for (m, n) in ((20000 + 37, 256), (65536, 256), (30001, 192)):
    ld = m + 3
    A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5           # column-major m x n, ld = m + 3
    Q = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5)  # logical upper T (row-major tensor)
    Tcm = T.T.contiguous()                                                       # column-major storage of T
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), ld, 0.0, capi.ptr(G), n)
    h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(Tcm), n, capi.ptr(A), ld, capi.ptr(Q), ld)
    h.sync()
    Am = A[:, :m].T                                                              # m x n logical
    Gref = torch.triu(Am.T @ Am)
    Gout = torch.triu(G.T)                                                       # G stored column-major -> logical = G.T
    Qref = Am @ T
    Qout = Q[:, :m].T
    eg = (Gout - Gref).abs().max().item() / Gref.abs().max().item()
    eq = (Qout - Qref).abs().max().item() / Qref.abs().max().item()
    print(f"m={m} n={n}: gram rel err {eg:.2e}  trmm-right rel err {eq:.2e}", flush=True)
    assert eg < 1e-13 and eq < 1e-13
print("ok")
