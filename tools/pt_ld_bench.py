"""capi_dpotrf_trtri(n) on a diagonal block INSIDE a larger matrix (leading dimension ld >> n), as the recursion calls it: does the
stride cost anything?   python tools/pt_ld_bench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
ms = C.c_float()
for n in (128, 1024, 2048, 4096):
    for ld in (n, 8192 + 0, 32768, 65536, 65536 + 16):
        if ld < n: continue
        cols = n
        big = torch.zeros((cols, ld), dtype=torch.float64, device="cuda")        # column-major n columns of height ld
        bigx = torch.zeros((cols, ld), dtype=torch.float64, device="cuda")
        A = torch.rand((n, n), dtype=torch.float64, device="cuda") - 0.5
        S = A @ A.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
        best = 1e9
        for _ in range(5):
            big[:, :n] = S; bigx.zero_(); torch.cuda.synchronize()
            h.call("capi_timer_start"); h.call("capi_dpotrf_trtri", n, capi.ptr(big), ld, capi.ptr(bigx), ld); h.call("capi_timer_stop_ms", C.byref(ms))
            best = min(best, ms.value)
        print(f"n={n} ld={ld}: {best*1e3:.0f} us", flush=True)
