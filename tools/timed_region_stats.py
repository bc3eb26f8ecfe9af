"""From a rocprofv3 kernel_trace.csv of `python bench.py --steps K --warmup W`: average duration of one kernel symbol over
the launches that fall inside bench.py's timed region (the last K factor() calls), for comparison with the HIP-event
average bench.py prints in roofline.avg_launch_ms.   usage: timed_region_stats.py trace.csv K [symbol-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]); sym = sys.argv[3] if len(sys.argv) > 3 else "dgemm_tile_kernel<128, true, true>"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a factor() call opens with lacpy_kernel (upper-triangle copy of the input's leading block) followed by leaf kernels
starts = [i for i, r in enumerate(rows) if "lacpy_kernel" in r["Kernel_Name"]]
calls = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    if sum("leaf128" in r["Kernel_Name"] for r in rows[a:b]) >= 8:
        seg = rows[a:b]
        ser = [i for i, r in enumerate(seg) if "serialize_kernel" in r["Kernel_Name"]]
        calls.append(seg[: ser[4] + 1] if len(ser) >= 5 else seg)       # a call ends with its 3 early + 2 late packing kernels
timed = [r for seg in calls[-K:] for r in seg if sym in r["Kernel_Name"]]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
print(f"{sym}: {len(d)} launches in the last {K} factor() calls, {len(d) / K:.1f} per call, average {sum(d) / len(d) / 1e6:.4f} ms, max {max(d) / 1e6:.3f} ms")
