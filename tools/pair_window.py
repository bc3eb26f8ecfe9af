"""Where do tile pairs beat the longest-first order?  dtrmm_oop at (ntri, nfree) shapes, CAPI_TRMM_PAIR=0/1 (run twice)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
def timeit(fn, reps=7):
    fn(); h.sync()
    ms = C.c_float(); best = 1e9
    for _ in range(reps):
        h.call("capi_timer_start"); fn(); h.call("capi_timer_stop_ms", C.byref(ms)); best = min(best, ms.value)
    return best
for ntri, nfree in ((2048, 8192), (2048, 16384), (4096, 2048), (4096, 4096), (4096, 8192), (4096, 16384), (6144, 6144), (8192, 4096), (8192, 8192), (8192, 16384), (12288, 12288)):
    T = torch.rand((ntri, ntri), dtype=torch.float64, device="cuda") - 0.5
    B = torch.rand((nfree, ntri), dtype=torch.float64, device="cuda") - 0.5
    Cm = torch.zeros((nfree, ntri), dtype=torch.float64, device="cuda")
    out = []
    for side, trans in ((0, 1), (0, 0), (1, 0)):
        m, n = (ntri, nfree) if side == 0 else (nfree, ntri)
        ms = timeit(lambda: h.call("capi_dtrmm_oop", side, 1, trans, 0, m, n, 1.0, capi.ptr(T), ntri, capi.ptr(B), m, capi.ptr(Cm), m))
        out.append(f"{'LR'[side]}{'NT'[trans]} {ms:8.3f} ms {ntri * ntri * nfree / ms / 1e9:6.2f}")
    print(f"pair={os.environ.get('CAPI_TRMM_PAIR', '1')} tri {ntri:6d} free {nfree:6d} wgs {ntri // 256 * (nfree // 128):6d} | " + " | ".join(out), flush=True)
