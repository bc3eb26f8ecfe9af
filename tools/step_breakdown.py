"""Per-step kernel breakdown of a rocprofv3 kernel trace of bench.py (last factor() call)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a factor() call starts with the upper-triangle copy of the input's leading block (lacpy_kernel) followed by leaves
starts = [i for i, r in enumerate(rows) if 'lacpy_kernel' in r['Kernel_Name']]
segs = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    if sum('leaf128' in r['Kernel_Name'] for r in rows[a:b]) >= 8: segs.append((a, b))
# merge consecutive lacpy starts belonging to the same call (head + rest copies)
a, b = segs[-1]
seg = rows[a:b]
ser = [i for i, r in enumerate(seg) if 'serialize_kernel' in r['Kernel_Name']]
if ser: seg = seg[:ser[-1] + 1] if len(ser) <= 6 else seg
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
