"""Device copy / fill / read rates of this box at CholeskyQR2's panel size (context for the tall-skinny kernels)."""
import torch, time
n = 1 << 30   # 8 GiB of doubles
a = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
tc = t(lambda: b.copy_(a)); tf = t(lambda: b.zero_()); tr = t(lambda: a.sum())
print(f"copy 8 GiB: {tc*1e3:.2f} ms ({2*8*n/tc/1e12:.2f} TB/s r+w)  fill: {tf*1e3:.2f} ms ({8*n/tf/1e12:.2f} TB/s)  read(sum): {tr*1e3:.2f} ms ({8*n/tr/1e12:.2f} TB/s)")
