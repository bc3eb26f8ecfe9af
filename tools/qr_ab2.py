"""CholeskyQR2 (m = 2^22, n = 256) end to end with the panel32 intermediate switched on and off from one factor() to the next in ONE
process (qr::cacqr reads CAPITAL_NO_PANEL32 per call): same box, same clocks, same buffers.   python tools/qr_ab2.py [log2_m] [pairs]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capital_amd import driver
driver.init(0, 0, 1, None, use_torch_stream=False)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m, n = 1 << lg, 256
q = driver.Cacqr(m, n, c=1, variant=2)
q.generate()
t = {"panel32": [], "column-major": []}
for i in range(2 * pairs + 4):
    off = i & 1
    if off:
        os.environ["CAPITAL_NO_PANEL32"] = "1"
    else:
        os.environ.pop("CAPITAL_NO_PANEL32", None)
    driver.sync()
    t0 = time.perf_counter(); q.factor(); driver.sync(); dt = (time.perf_counter() - t0) * 1e3
    if i >= 4:
        t["column-major" if off else "panel32"].append(dt)
os.environ.pop("CAPITAL_NO_PANEL32", None)
tag = os.path.basename(os.environ.get("CAPITAL_HIP_LIB", "libcapital_hip.so"))
for k, v in t.items():
    print(f"[{tag}] cacqr2 m=2^{lg} n={n} Q1 {k:13s}: min {min(v):.3f}  median {statistics.median(v):.3f}  mean {statistics.fmean(v):.3f} ms  ({4.0 * m * n * n / statistics.median(v) / 1e9:.2f} TF/s at the median)", flush=True)
print(f"residual {q.residual():.2e} orth {q.orthogonality():.2e}", flush=True)
q.close()
driver.finalize()
