import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from capital_amd import capi, driver
driver.init(0, 0, 1, None, use_torch_stream=False)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
q = driver.Cacqr(m, n, c=1, variant=2)
q.generate()
for _ in range(3):
    q.factor()
driver.sync()
print("residual", q.residual(), "orth", q.orthogonality())
q.close()
driver.finalize()
