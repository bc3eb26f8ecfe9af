"""CholeskyQR2 end to end, config 3 (m = 2^22, n = 256) and a slice of config 5: ms per factor(), residual, orthogonality.
A/B switches are environment variables read by the library per call or per process (run one process per setting):
  CAPITAL_NO_PANEL32=1   column-major intermediate Q1 (round 2)
usage: python tools/qr_ab.py [log2_m] [n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from capital_amd import driver
driver.init(0, 0, 1, None, use_torch_stream=False)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m = 1 << lg
q = driver.Cacqr(m, n, c=1, variant=2)
q.generate()
for _ in range(2):
    q.factor()
driver.sync()
best, tot = 1e9, 0.0
for _ in range(reps):
    t0 = time.perf_counter(); q.factor(); driver.sync(); dt = (time.perf_counter() - t0) * 1e3
    best = min(best, dt); tot += dt
flops = 4.0 * m * n * n
tag = " ".join(f"{k}={os.environ[k]}" for k in ("CAPITAL_NO_PANEL32",) if k in os.environ) or "default"
print(f"cacqr2 m=2^{lg} n={n} [{tag}]: best {best:.3f} ms ({flops / best / 1e9:.2f} TF/s)  mean {tot / reps:.3f} ms  residual {q.residual():.2e} orth {q.orthogonality():.2e}", flush=True)
q.close()
driver.finalize()
