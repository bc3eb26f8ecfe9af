"""Can two ranks share ONE GPU under RCCL on this box?  (functional rehearsal of the N=2 path on a 1-GPU machine)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from capital_amd import driver
import numpy as np
try:
    driver.init_distributed(0)
    print(rank, "rccl world ok", flush=True)
    p = driver.Cholinv(512, c=2 if world == 2 else 1, complete_inv=1, bc_mult=-2, serialize=False, bc_policy=0)
    p.generate(); p.factor()
    print(rank, "residual", p.residual(), flush=True)
    p.close()
    driver.finalize()
except Exception as e:
    print(rank, "FAILED:", repr(e)[:500], flush=True)
dist.destroy_process_group()
