"""The two tall-skinny kernels of CholeskyQR2 at n = 256, one by one, column-major and panel32 forms, HIP-event timed in ONE process
(rounds interleaved so that clock drift hits every form alike): min and median of `reps` launches each.
  python tools/ts_ab.py [log2_m] [reps]      A/B of builds: CAPITAL_HIP_LIB=<other libcapital_hip.so>"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m, n = 1 << lg, 256
h = capi.Handle(0)
torch.manual_seed(0)
A = torch.rand((n, m), dtype=torch.float64, device="cuda")
A32 = A.T.reshape(m // 32, 32, n).transpose(1, 2).contiguous()
Q = torch.empty_like(A)
G = torch.zeros((n, n), dtype=torch.float64, device="cuda")
T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda")).T.contiguous()
has32 = hasattr(capi.load(), "capi_dtrmm_right_panel32")
forms = {
    "gram cm": lambda: h.call("capi_dsyrk", 1, 1, n, m, 1.0, capi.ptr(A), m, 0.0, capi.ptr(G), n),
    "trmm cm->cm": lambda: h.call("capi_dtrmm_oop", 1, 1, 0, 0, m, n, 1.0, capi.ptr(T), n, capi.ptr(A), m, capi.ptr(Q), m),
}
if has32:
    forms["gram p32"] = lambda: h.call("capi_dsyrk_panel32", n, m, 1.0, capi.ptr(A32), 0.0, capi.ptr(G), n)
    forms["trmm cm->p32"] = lambda: h.call("capi_dtrmm_right_panel32", m, n, 1.0, capi.ptr(T), n, capi.ptr(A), m, capi.ptr(Q), 0)
    forms["trmm p32->cm"] = lambda: h.call("capi_dtrmm_right_panel32", m, n, 1.0, capi.ptr(T), n, capi.ptr(A32), 0, capi.ptr(Q), m)
    forms["trmm p32->p32"] = lambda: h.call("capi_dtrmm_right_panel32", m, n, 1.0, capi.ptr(T), n, capi.ptr(A32), 0, capi.ptr(Q), 0)
times = {k: [] for k in forms}
ms = C.c_float()
for k, f in forms.items():
    f()
h.sync()
for _ in range(reps):
    for k, f in forms.items():
        h.call("capi_timer_start"); f(); h.call("capi_timer_stop_ms", C.byref(ms))
        times[k].append(ms.value)
flops = float(m) * n * n
tag = os.path.basename(os.environ.get("CAPITAL_HIP_LIB", "libcapital_hip.so"))
for k, v in times.items():
    print(f"[{tag}] {k:14s} m=2^{lg}: min {min(v):.3f} ms ({flops / min(v) / 1e9:.1f} TF/s)  median {statistics.median(v):.3f} ms", flush=True)
