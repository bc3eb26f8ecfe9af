#!/bin/bash
# Round-4 evidence, one gpurun call.  Every leg reports its own exit code into $O/legs.txt; a required leg that fails makes the script exit
# non-zero at the end.  Outputs: gpurun_out/cap4/ (what is to be kept is copied into profiles/ as r4_*).
#   1  the bench line (launches in resident rounds are the default now; roofline from the kernels' own interval stamps)
#   2  the same command under rocprofv3 --kernel-trace --stats: per-symbol averages and the union of the intervals, cut to the timed region,
#      for BOTH symbols of the roofline's kernel (dgemm_tile_kernel<128,true,true>, dtrmm_pair_kernel<true,true>); idle analysis
#   3  PMC passes of the headline step (separate runs per counter set: FETCH_SIZE; WRITE_SIZE; MFMA busy + clock), both symbols
#   4  PMC pass of the TRSM-mode leg in a process of its own
#   5  PMC passes of the tall-skinny kernels: n = 256 (column-major and panel32 forms) and n = 1024 (config 5's width)
#   6  kernel trace of two n = 32768 steps: idle analysis of config 2
#   7  the whole bench process (config 2 and both TRSM legs included) under --pmc: rounds 2 and 3 recorded a host SIGSEGV here; located in round 4
#      (a queue-ring wrap inside the profiler's packet interceptor, r4_pmc_whole_bench_crash.txt) and avoided by a ring that never wraps: required to pass
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/cap4
rm -rf $O; mkdir -p $O
fail=0
leg() { echo "leg $1 rc=$2" | tee -a $O/legs.txt; if [ "$2" -ne 0 ]; then fail=1; fi; }
TILE='dgemm_tile_kernel<128, true, true>'
PAIR='dtrmm_pair_kernel<true, true>'

timeout -k 10 900 python bench.py --steps 3 > $O/bench.json 2> $O/bench.err; leg 1-bench $?

timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/rp.err; rc=$?
if [ $rc -eq 0 ]; then
  F=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
  { python tools/timed_region_stats.py $F 3 1 "$TILE"; python tools/timed_region_stats.py $F 3 1 "$PAIR"; python tools/timed_region_stats.py $F 3 1 "$TILE|$PAIR"; } > $O/timed_region.txt; rc=$?
  cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python tools/gap_analysis.py $F 4 > $O/gaps_n65536.txt 2>&1 || true        # W + K = 4 factor() calls of the headline
fi
rm -rf $O/tr; leg 2-kernel-trace $rc

for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -c1-2 | tr A-Z a-z)
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$t -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr --no-config2 > $O/pmc_$t.json 2> $O/pmc_$t.err; rc=$?
  if [ $rc -eq 0 ]; then
    grep -E "Counter_Name|dgemm_tile_kernel<128, true, true>|dtrmm_pair_kernel<true, true>" $(find $O/pmc_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_all.csv
    python - <<PY
import csv, json
j = json.loads([l for l in open('$O/pmc_$t.json') if l.startswith('{')][-1])
want = {s["symbol"]: int(round(s["launches_per_step"])) for s in j["roofline"]["symbols"]}      # launches of the ONE factor() (the validator's products follow)
c = 2 if '$t' == 'sq' else 1
rows = list(csv.reader(open('$O/pmc_${t}_all.csv')))
hdr, body = rows[0], rows[1:]
ki = hdr.index("Kernel_Name")
seen = {k: 0 for k in want}
keep = []
for r in body:
    for k in want:
        if k in r[ki] and seen[k] < want[k] * c:
            seen[k] += 1; keep.append(r); break
with open('$O/pmc_${t}_bench_n65536.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(hdr); w.writerows(keep)
print('pmc $t: kept', seen, 'of', {k: v * c for k, v in want.items()})
PY
    rc=$?
  fi
  rm -rf $O/pmc_$t $O/pmc_${t}_all.csv; leg 3-pmc-$t $rc
done

timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_trsm -o p -- python tools/pmc_segv_probe.py 65536 1 > $O/pmc_trsm.out 2> $O/pmc_trsm.err; rc=$?
if [ $rc -eq 0 ]; then
  python - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$(find $O/pmc_trsm -name 'p_counter_collection.csv' | head -1)")) if r["Counter_Name"] == "FETCH_SIZE"]
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
    tot[k] += float(r["Counter_Value"]) * 1024 * 2 / 1e9; cnt[k] += 1
with open("$O/pmc_fe_trsm_mode_n65536.txt", "w") as f:
    f.write("TRSM-mode factor() at n = 65536 under rocprofv3 --pmc FETCH_SIZE, one process (tools/pmc_segv_probe.py 65536 1): 2 x FETCH_SIZE per kernel symbol, GB (launches)\n")
    for k, v in tot.most_common(12):
        f.write(f"{v:10.1f}  ({cnt[k]:6d})  {k}\n")
    f.write(f"{sum(tot.values()):10.1f}  ({sum(cnt.values()):6d})  all kernels\n")
PY
  rc=$?
fi
rm -rf $O/pmc_trsm; leg 4-pmc-trsm-mode $rc

for c in FETCH_SIZE WRITE_SIZE; do
  t=$(echo $c | cut -c1-2 | tr A-Z a-z)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ts_$t -o p -- python tools/ts_ab.py 22 2 > $O/ts256_$t.log 2>&1; rc=$?
  if [ $rc -eq 0 ]; then grep -E "Counter_Name|gram_ts_kernel|trmm_right_ts32_kernel" $(find $O/ts_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_ts256.csv; rc=$?; fi
  rm -rf $O/ts_$t; leg 5-pmc-ts256-$t $rc
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tw_$t -o p -- python tools/ts_wide_bench.py 21 > $O/ts1024_$t.log 2>&1; rc=$?
  if [ $rc -eq 0 ]; then grep -E "Counter_Name|gram_ts_kernel|dgemm_tile_kernel" $(find $O/tw_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_ts1024.csv; rc=$?; fi
  rm -rf $O/tw_$t; leg 5-pmc-ts1024-$t $rc
done

timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr2 -o b -- python bench.py --n 32768 --steps 2 --no-cpu --no-qr > $O/bench_n32768_under_rocprof.json 2> $O/rp2.err; rc=$?
if [ $rc -eq 0 ]; then python tools/gap_analysis.py $(find $O/tr2 -name "b_kernel_trace.csv" | head -1) 3 > $O/gaps_n32768.txt 2>&1; rc=$?; fi
rm -rf $O/tr2; leg 6-gaps-n32768 $rc

# The crash case of rounds 2 and 3 (host SIGSEGV of the whole bench process under --pmc), located and resolved in round 4
# (profiles/r4_pmc_whole_bench_crash.txt: the profiler's packet interceptor reads past the end of a 16384-packet queue ring; bench.py now asks for a ring
# that never wraps whenever a rocprofiler tool is attached).  REQUIRED to pass from now on.
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_whole -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_whole.json 2> $O/pmc_whole.err; rc=$?
if [ $rc -eq 0 ]; then python - > $O/pmc_whole_dispatches.txt <<PY
import csv, collections
f = "$(find $O/pmc_whole -name 'p_counter_collection.csv' | head -1)"
cnt = collections.Counter(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:] for r in csv.DictReader(open(f)))
print("whole bench process under rocprofv3 --pmc FETCH_SIZE: dispatches profiled:", sum(cnt.values()))
for k, v in cnt.most_common(16): print("  ", v, k)
PY
else tail -60 $O/pmc_whole.err > $O/pmc_whole_tail.err; fi
rm -rf $O/pmc_whole $O/pmc_whole.err; leg 7-whole-bench-under-pmc $rc

ls -la $O; cat $O/legs.txt; cat $O/timed_region.txt; cut -c1-400 $O/bench.json
exit $fail
