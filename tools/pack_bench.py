"""Time of packing the upper triangle of an n x n block (capi_serialize_shape rect -> uppertri), the copy cholinv runs last."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
A = torch.rand((n, n), dtype=torch.float64, device="cuda")
P = torch.zeros(n * (n + 1) // 2, dtype=torch.float64, device="cuda")
args = (capi.UPPERTRI, capi.RECT, capi.UPPERTRI, capi.ptr(A), n, n, capi.ptr(P), n, n, 0, n, 0, n, 0, n, 0, n)
ms = C.c_float(); best = 1e9
for _ in range(6):
    h.call("capi_timer_start"); h.call("capi_serialize_shape", *args); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
At = A.T                                             # logical column-major view: At[i, j] = element (row i, column j)
iu = torch.triu_indices(n, n, device="cuda")
ok = True
for j in (0, 1, 2, 255, 256, 1023, 1024, n - 1):
    off = j * (j + 1) // 2
    ok &= bool(torch.equal(P[off:off + j + 1], At[:j + 1, j]))
print(f"pack upper {n}: {best:.3f} ms ({2 * 8 * n * (n + 1) / 2 / best / 1e9:.2f} TB/s r+w), spot check {'ok' if ok else 'MISMATCH'}")
