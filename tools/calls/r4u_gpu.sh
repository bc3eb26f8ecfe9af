#!/bin/bash
# round 4, call u: per-kernel durations of the blocked diagonal-block routine, panel by strip solve against panel by TRMM
export TMPDIR=/tmp
O=gpurun_out/r4u; rm -rf $O; mkdir -p $O
for v in 1 0; do
  CAPI_PANEL_SOLVE=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$v -o p -- python tools/pt_bench.py > $O/pt$v.txt 2>&1
  cp $(find $O/tr$v -name "p_kernel_stats.csv" | head -1) $O/kernel_stats_solve$v.csv; rm -rf $O/tr$v
  echo "== CAPI_PANEL_SOLVE=$v"; python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_solve$v.csv")))
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:9]:
    print(f"{int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:8.1f}   {r['Name'].replace('(anonymous namespace)::','')[:60]}")
PY
done
