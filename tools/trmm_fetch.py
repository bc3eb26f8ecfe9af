"""The recursion's big triangular products, R12 = R11^-T A12 (Left / Upper / Trans) at orders 16384 and 32768, alone: time per launch
(HIP events) -- and, under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`, what they fetch.  The launch form is chosen by
the environment: CAPI_TRMM_PAIR=2 (tile pairs whenever the launch is whole resident rounds), CAPI_TRMM_PAIR_ROUNDS=1 (one launch per round).
  python tools/trmm_fetch.py [order ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
orders = [int(a) for a in sys.argv[1:]] or [16384, 32768]
h = capi.Handle(0)
tag = " ".join(f"{k}={os.environ[k]}" for k in ("CAPI_TRMM_PAIR", "CAPI_TRMM_PAIR_ROUNDS") if k in os.environ) or "default"
for n in orders:
    torch.manual_seed(n)
    T = torch.triu(torch.rand((n, n), dtype=torch.float64, device="cuda")).T.contiguous()      # column-major upper
    B = torch.rand((n, n), dtype=torch.float64, device="cuda")
    Cc = torch.empty_like(B)
    f = lambda: h.call("capi_dtrmm_oop", 0, 1, 1, 0, n, n, 1.0, capi.ptr(T), n, capi.ptr(B), n, capi.ptr(Cc), n)
    f(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(3):
        h.call("capi_timer_start"); f(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    chk = ""
    if n <= 16384:
        ref = T.T.T @ B.T.T if False else None
    print(f"[{tag}] trmm L/U/T order {n}: {best:.2f} ms  {float(n) ** 3 / best / 1e9:.2f} TF/s", flush=True)
    del T, B, Cc
    torch.cuda.empty_cache()
