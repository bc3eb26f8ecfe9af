"""Device memory held by a cholinv problem under SaveIntermediates / FlushIntermediates (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import driver
driver.init(0, 0, 1, None, use_torch_stream=False)
n = 4096
blk = 8 * n * n / 2**20
def free(): 
    driver.sync(); return torch.cuda.mem_get_info()[0] / 2**20
for flush in (False, True, False, True):
    f0 = free()
    p = driver.Cholinv(n, bc_mult=-2, serialize=True, flush_intermediates=flush)
    f1 = free(); p.generate(); p.factor(); f2 = free()
    p.factor(); f3 = free()
    r = p.residual(); f4 = free()
    p.close(); f5 = free()
    print(f"flush={flush}: block {blk:.0f} MiB | create {f0-f1:.0f} | after factor {f0-f2:.0f} | after 2nd factor {f0-f3:.0f} | after residual {f0-f4:.0f} | after close {f0-f5:.0f}  (residual {r:.1e})", flush=True)
driver.finalize()
