"""Per-step kernel breakdown of a rocprofv3 kernel trace of bench.py (last factor() call).
usage: step_breakdown.py trace.csv [number of factor() calls = warmup + steps]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a factor() call issues exactly five packing kernels (serialize_kernel), so call i ends with the 5(i+1)-th of the process;
# argv[2] = number of factor() calls in the trace (warm-up + timed, default 4): the last one is broken down
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ser = [i for i, r in enumerate(rows) if 'serialize_kernel' in r['Kernel_Name']]
assert len(ser) >= 5 * ncalls, "fewer packing kernels than factor() calls"
seg = rows[ser[5 * (ncalls - 1) - 1] + 1: ser[5 * ncalls - 1] + 1]
t0 = int(seg[0]['Start_Timestamp']); end = max(int(r['End_Timestamp']) for r in seg)
agg = defaultdict(lambda: [0, 0])
for r in seg:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:52]
    if 'dgemm_tile' in nm:
        d = en - st
        nm += ' >1ms' if d > 1e6 else (' 0.1-1ms' if d > 1e5 else ' <0.1ms')
    agg[nm][0] += 1; agg[nm][1] += en - st
print(f"span {(end - t0) / 1e6:.1f} ms, sum of kernels {sum(v[1] for v in agg.values()) / 1e6:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / 1e6:8.2f} ms {v[0]:5d} x {v[1] / v[0] / 1e3:9.1f} us  {k}")
if len(sys.argv) > 3:      # argv[3] = kernel-name substring: list those launches with start offset and duration
    for r in seg:
        if sys.argv[3] in r['Kernel_Name']:
            print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} ms  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {r['Kernel_Name'][:40]}")
