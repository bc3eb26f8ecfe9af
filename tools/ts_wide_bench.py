"""Tall-skinny Gram matrix and Q = A T at widths above 256 (CholeskyQR2 config 5: n = 1024): correctness against torch fp64
on a ragged shape, then timings.   python tools/ts_wide_bench.py [log2_m ...] [--n 1024]   (A/B: CAPI_NO_TALL=1)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi

args = [a for a in sys.argv[1:]]
n = 1024
if "--n" in args:
    i = args.index("--n"); n = int(args[i + 1]); del args[i:i + 2]
logs = [int(a) for a in args] or [21]
h = capi.Handle(0)
torch.manual_seed(1)


def check(m, n):
    ld = m + 2
    A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5           # column-major m x n, ld = m + 2
    Q = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5)
    Tcm = T.T.contiguous()
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), ld, 0.0, capi.ptr(G), n)
    h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(Tcm), n, capi.ptr(A), ld, capi.ptr(Q), ld)
    h.sync()
    Am = A[:, :m].T
    Gref = torch.triu(Am.T @ Am)
    eg = (torch.triu(G.T) - Gref).abs().max().item() / Gref.abs().max().item()
    Qref = Am @ T
    eq = (Q[:, :m].T - Qref).abs().max().item() / Qref.abs().max().item()
    print(f"check m={m} n={n}: gram rel err {eg:.2e}  trmm-right rel err {eq:.2e}", flush=True)
    assert eg < 1e-13 and eq < 1e-13


check(128 * 64 * (n // 128) + 77, n)           # tall enough for the tall ordering, ragged last row tile
check((1 << 21) + 5, n)                        # tall enough for the sliced Gram matrix (K / 32768 * tiles >= 2048 at n = 1024)


def timeit(fn, reps=3):
    fn(); h.sync()
    ms = C.c_float()
    best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms))
        best = min(best, ms.value)
    return best


for lg in logs:
    m = 1 << lg
    A = torch.rand((n, m), dtype=torch.float64, device="cuda")
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    Q = torch.empty_like(A)
    ms = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), m, 0.0, capi.ptr(G), n))
    print(f"gram m=2^{lg} n={n}: {ms:.3f} ms {m*n*n/ms/1e9:.2f} TF/s  {8*m*n/ms/1e6:.0f} GB/s algorithmic", flush=True)
    ms = timeit(lambda: h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(G), n, capi.ptr(A), m, capi.ptr(Q), m))
    print(f"trmm-right m=2^{lg} n={n}: {ms:.3f} ms {m*n*n/ms/1e9:.2f} TF/s  {16*m*n/ms/1e6:.0f} GB/s algorithmic", flush=True)
    del A, G, Q
    torch.cuda.empty_cache()

if os.environ.get("TSW_EXTRA"):
    # ceilings of the tile kernel on the same tall shape: a plain product (no triangle) at K = n and at K = 128
    m = 1 << 21
    A = torch.rand((n, m), dtype=torch.float64, device="cuda")
    Q = torch.empty_like(A)
    for k in (n, 128):
        B = torch.rand((n, k), dtype=torch.float64, device="cuda")
        ms = timeit(lambda: h.call("capi_dgemm", 0, 0, m, n, k, 1.0, capi.ptr(A), m, capi.ptr(B), k, 0.0, capi.ptr(Q), m))
        print(f"dgemm NN m=2^21 n={n} k={k}: {ms:.3f} ms {2*m*n*k/ms/1e9:.2f} TF/s", flush=True)
