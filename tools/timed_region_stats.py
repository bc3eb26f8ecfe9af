"""From a rocprofv3 kernel_trace.csv of `python bench.py --steps K --warmup W`: average duration of one kernel symbol over
the launches that fall inside bench.py's timed region (the last K of the first W+K factor() calls), for comparison with
the HIP-event average bench.py prints in roofline.avg_launch_ms.
usage: timed_region_stats.py trace.csv K [W=1] [symbol-substring[|another]]
A factor() call issues exactly five packing kernels (serialize_kernel: three beside the top-level update, two at its end),
so call i ends with the 5(i+1)-th serialize_kernel of the process."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]); W = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sym = sys.argv[4] if len(sys.argv) > 4 else "dgemm_tile_kernel<128, true, true>"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ser = [i for i, r in enumerate(rows) if "serialize_kernel" in r["Kernel_Name"]]
assert len(ser) >= 5 * (W + K), "fewer packing kernels than W+K factor() calls"
ends = [ser[5 * (c + 1) - 1] for c in range(W + K)]
a, b = ends[W - 1] + 1 if W > 0 else 0, ends[W + K - 1]
timed = [r for r in rows[a:b + 1] if any(x in r["Kernel_Name"] for x in sym.split("|"))]      # "A|B": the launches of either symbol, one union
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed]
print(f"{sym}: {len(d)} launches in the {K} timed factor() calls, {len(d) / K:.1f} per call, average {sum(d) / len(d) / 1e6:.4f} ms, max {max(d) / 1e6:.3f} ms")
# union of the launches' execution intervals (what bench.py's roofline.union_ms_per_step is, from the kernels' own stamps): launches of
# several streams interleave round by round, so the sum of the durations exceeds the time the kernel was on the device
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in timed)
cov, ca, cb = 0, None, None
for s_, e_ in iv:
    if ca is None or s_ > cb:
        if ca is not None:
            cov += cb - ca
        ca, cb = s_, e_
    elif e_ > cb:
        cb = e_
cov += (cb - ca) if ca is not None else 0
print(f"{sym}: per call: union of the intervals {cov / K / 1e6:.3f} ms, sum of the durations {sum(d) / K / 1e6:.3f} ms")
