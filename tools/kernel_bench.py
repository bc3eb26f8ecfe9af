"""Per-kernel timing on the GPU box (not the judged bench): fp64 MFMA peak loop, GEMM/GEMMT/TRMM tile kernel."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from capital_amd import capi

h = capi.Handle(0)
out = {}
v = C.c_double()
for it in (20000, 100000):
    h.call("capi_mfma_f64_peak", it, C.byref(v))
out["mfma_f64_peak_tflops"] = v.value
print("mfma peak TF/s", v.value, flush=True)


def timeit(fn, reps=5):
    fn(); h.sync()
    ms = C.c_float()
    best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms))
        best = min(best, ms.value)
    return best


sizes = [int(s) for s in sys.argv[1:]] or [4096, 8192]
for n in sizes:
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    for name, ta, tb in (("TN", 1, 0), ("NN", 0, 0), ("NT", 0, 1), ("TT", 1, 1)):
        ms = timeit(lambda: h.call("capi_dgemm", ta, tb, n, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, 0.0, capi.ptr(Cm), n))
        tf = 2.0 * n ** 3 / ms / 1e9
        out[f"dgemm_{name}_{n}"] = tf
        print(f"dgemm {name} n={n}: {ms:.3f} ms  {tf:.2f} TF/s", flush=True)
    ms = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, n, -1.0, capi.ptr(A), n, 1.0, capi.ptr(Cm), n))
    tf = 1.0 * n ** 3 / ms / 1e9
    out[f"dsyrk_UT_{n}"] = tf
    print(f"dsyrk U/T n=k={n}: {ms:.3f} ms  {tf:.2f} TF/s (n^2 k flops)", flush=True)
    ms = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(A), n, capi.ptr(B), n, capi.ptr(Cm), n))
    tf = 1.0 * n ** 3 / ms / 1e9
    out[f"dtrmm_LUT_{n}"] = tf
    print(f"dtrmm L/U/T n={n}: {ms:.3f} ms  {tf:.2f} TF/s (n^3 flops)", flush=True)
    X = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    W = S.clone()
    def f():
        W.copy_(S)
        h.call("capi_dpotrf_trtri", n, capi.ptr(W), n, capi.ptr(X), n)
    ms = timeit(f, 3)
    print(f"potrf_trtri n={n}: {ms:.3f} ms  {n**3/3/ms/1e9:.2f} TF/s (n^3/3)", flush=True)
    out[f"potrf_trtri_{n}"] = n ** 3 / 3 / ms / 1e9
# tall-skinny (CholeskyQR2 shapes)
for m, n in ((1 << 20, 256), (1 << 18, 1024)):
    A = torch.rand((n, m), dtype=torch.float64, device="cuda")
    G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    Q = torch.empty_like(A)
    ms = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), m, 0.0, capi.ptr(G), n))
    print(f"gram m={m} n={n}: {ms:.3f} ms {m*n*n/ms/1e9:.2f} TF/s  {8*m*n/ms/1e6:.0f} GB/s", flush=True)
    out[f"gram_{m}_{n}"] = m * n * n / ms / 1e9
    ms = timeit(lambda: h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(G), n, capi.ptr(A), m, capi.ptr(Q), m))
    print(f"trmm-right m={m} n={n}: {ms:.3f} ms {m*n*n/ms/1e9:.2f} TF/s  {16*m*n/ms/1e6:.0f} GB/s", flush=True)
    out[f"trmmR_{m}_{n}"] = m * n * n / ms / 1e9
print(json.dumps(out))
