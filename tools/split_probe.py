"""Does a 32768-order triangular product run faster cut into 16384-order launches?  (tile-order probe, not the judged bench)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi

h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
hn = n // 2
ld = int(sys.argv[2]) if len(sys.argv) > 2 else n        # leading dimension of all three matrices
A = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
B = torch.rand((n, ld), dtype=torch.float64, device="cuda") - 0.5
Cm = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
pA, pB, pC = capi.ptr(A), capi.ptr(B), capi.ptr(Cm)


def off(p, i, j):            # column-major element (i, j), ld n
    return p + 8 * (i + j * ld)


def timeit(fn, reps=3):
    fn(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms))
        best = min(best, ms.value)
    return best


ms = timeit(lambda: h.call("capi_dsyrk", 1, 1, n, n, -1.0, pA, ld, 1.0, pC, ld))
print(f"dsyrk U/T {n}: one launch {ms:.2f} ms {n**3/ms/1e9:.2f} TF/s", flush=True)


def syrk_split():
    # C11 -= A1^T A1, C22 -= A2^T A2, C12 -= A1^T A2   (A = [A1 A2], column blocks; k = n rows)
    h.call("capi_dsyrk", 1, 1, hn, n, -1.0, pA, ld, 1.0, pC, ld)
    h.call("capi_dgemm", 1, 0, hn, hn, n, -1.0, pA, ld, off(pA, 0, hn), ld, 1.0, off(pC, 0, hn), ld)
    h.call("capi_dsyrk", 1, 1, hn, n, -1.0, off(pA, 0, hn), ld, 1.0, off(pC, hn, hn), ld)


ms = timeit(syrk_split)
print(f"dsyrk U/T {n}: syrk+gemm+syrk {ms:.2f} ms {n**3/ms/1e9:.2f} TF/s", flush=True)
for nm, f in (("syrk half", lambda: h.call("capi_dsyrk", 1, 1, hn, n, -1.0, pA, ld, 1.0, pC, ld)),
              ("gemm half", lambda: h.call("capi_dgemm", 1, 0, hn, hn, n, -1.0, pA, ld, off(pA, 0, hn), ld, 1.0, off(pC, 0, hn), ld))):
    print(f"   {nm}: {timeit(f):.2f} ms", flush=True)

# TRMM  C = T^T B, T upper (left, upper, trans, non-unit)
ms = timeit(lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, pA, ld, pB, ld, pC, ld))
print(f"dtrmm L/U/T {n}: one launch {ms:.2f} ms {n**3/ms/1e9:.2f} TF/s", flush=True)


def trmm_split():
    # C1 = T11^T B1 ; C2 = T22^T B2 + T12^T B1
    h.call("capi_dtrmm_oop", 0, 1, 1, 0, hn, n, 1.0, pA, ld, pB, ld, pC, ld)
    h.call("capi_dtrmm_oop", 0, 1, 1, 0, hn, n, 1.0, off(pA, hn, hn), ld, off(pB, hn, 0), ld, off(pC, hn, 0), ld)
    h.call("capi_dgemm", 1, 0, hn, n, hn, 1.0, off(pA, 0, hn), ld, pB, ld, 1.0, off(pC, hn, 0), ld)


ms = timeit(trmm_split)
print(f"dtrmm L/U/T {n}: trmm+trmm+gemm {ms:.2f} ms {n**3/ms/1e9:.2f} TF/s", flush=True)
for nm, f in (("trmm half", lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, hn, n, 1.0, pA, ld, pB, ld, pC, ld)),
              ("gemm half", lambda: h.call("capi_dgemm", 1, 0, hn, n, hn, 1.0, off(pA, 0, hn), ld, pB, ld, 1.0, off(pC, hn, 0), ld))):
    print(f"   {nm}: {timeit(f):.2f} ms", flush=True)
