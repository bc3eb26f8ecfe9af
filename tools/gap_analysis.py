"""Idle analysis of the last factor() call in a kernel trace: time during which NO kernel >= 0.5 ms is running, by what runs meanwhile."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ser = [i for i, r in enumerate(rows) if "serialize_kernel" in r["Kernel_Name"]]
ncall = int(sys.argv[2])            # factor() calls in the trace (5 packing kernels each)
a = ser[5 * (ncall - 1) - 1] + 1 if ncall > 1 else 0
b = ser[5 * ncall - 1]
seg = rows[a:b + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
big = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 500000]
big.sort()
# union of big intervals
cov = 0; cur_s, cur_e = None, None
gaps = []
for s, e in big:
    if cur_e is None: cur_s, cur_e = s, e; gaps.append((t0, s))
    elif s <= cur_e: cur_e = max(cur_e, e)
    else: cov += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
cov += cur_e - cur_s; gaps.append((cur_e, t1))
span = t1 - t0
print(f"span {span/1e6:.1f} ms, big kernels cover {cov/1e6:.1f} ms, uncovered {(span-cov)/1e6:.1f} ms in {len([g for g in gaps if g[1]>g[0]])} gaps")
gaps = sorted([(e - s, s, e) for s, e in gaps if e > s], reverse=True)
for d, s, e in gaps[:12]:
    names = collections.Counter()
    for r in seg:
        rs, re_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if rs < e and re_ > s and re_ - rs < 500000: names[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]] += (min(re_, e) - max(rs, s))
    top = ", ".join(f"{k} {v/1e6:.2f}" for k, v in names.most_common(4))
    print(f"  gap {d/1e6:6.2f} ms at +{(s-t0)/1e6:7.1f}: {top}")
