"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals of the LAST call segment (after the last non-capital kernel)."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'namespace' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
seg = rows[idx[-1] + 1:] if idx else rows
t0 = int(seg[0]['Start_Timestamp']); prev = t0; gaps = 0; tot = 0
agg = defaultdict(lambda: [0, 0])
for r in seg:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
    agg[nm][0] += 1; agg[nm][1] += en - st
    gaps += max(0, st - prev); prev = max(prev, en); tot += en - st
print(f"span {(prev - t0) / 1e3:.1f} us  busy {tot / 1e3:.1f}  gaps {gaps / 1e3:.1f}  launches {len(seg)}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1] / 1e3:9.1f} us {v[0]:5d} x {v[1] / v[0] / 1e3:8.2f}  {k}")
if len(sys.argv) > 2:
    for r in seg[:int(sys.argv[2])]:
        print(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44], round((int(r['Start_Timestamp']) - t0) / 1e3, 1),
              round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1), r['Grid_Size_X'], r['Workgroup_Size_X'])
