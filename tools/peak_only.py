import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi
h = capi.Handle(0)
v = C.c_double()
for it in (20000, 200000):
    h.call("capi_mfma_f64_peak", it, C.byref(v))
print("blocks/CU", os.environ.get("CAPI_PEAK_BLOCKS_PER_CU", "1"), "TF/s", v.value, flush=True)
