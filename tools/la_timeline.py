"""Timeline of the last factor() call of a traced bench.py: long kernels (>= 0.3 ms) with their queue, and for the short
kernels the busy time per 5 ms window per queue.  usage: la_timeline.py trace.csv [ncalls=4]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ser = [i for i, r in enumerate(rows) if 'serialize_kernel' in r['Kernel_Name']]
seg = rows[ser[5 * (ncalls - 1) - 1] + 1: ser[5 * ncalls - 1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in seg})
print("columns:", list(seg[0].keys()))
print("queues:", qs)
for r in seg:
    st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if en - st >= 300000:
        print(f"q{r['Queue_Id']:>3} +{st / 1e6:8.2f} .. {en / 1e6:8.2f}  {(en - st) / 1e6:7.2f} ms  {r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
win = defaultdict(lambda: [0, 0])
for r in seg:
    st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if en - st < 300000:
        w = win[(st // 5000000, r['Queue_Id'])]
        w[0] += 1; w[1] += en - st
print("short kernels per 5 ms window: (window start ms, queue): count, busy ms")
for k in sorted(win):
    print(f"  {k[0] * 5:4d} q{k[1]}: {win[k][0]:4d} {win[k][1] / 1e6:6.2f}")
