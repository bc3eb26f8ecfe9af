#!/bin/bash
# Round-3 evidence, one gpurun call.  Every leg reports its own exit code into $O/legs.txt; a leg that fails or is skipped makes the
# script exit non-zero at the end (round 2's capture silently depended on --no-config2 after a profiler crash).  Outputs: gpurun_out/cap3/
# (copy what is to be kept into profiles/ as r3_*).
#   1  the bench line                                   2  the same command under rocprofv3 --kernel-trace --marker-trace --stats
#   3  PMC passes of the headline step (separate runs per counter set: FETCH_SIZE; WRITE_SIZE; MFMA busy + clock)
#   4  PMC pass of the TRSM-mode leg in a process of its own (the leg that crashed the profiler's host side in round 2 after ~50 k
#      profiled dispatches of one process)               5  PMC passes of CholeskyQR2's tall-skinny kernels (column-major and panel32 forms)
#   6  ONE recorded attempt of the whole bench process (with config 2 and the TRSM legs) under --pmc: the round-2 crash case
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/cap3
rm -rf $O; mkdir -p $O
fail=0
leg() { echo "leg $1 rc=$2" | tee -a $O/legs.txt; if [ "$2" -ne 0 ]; then fail=1; fi; }

timeout -k 10 900 python bench.py --steps 3 > $O/bench.json 2> $O/bench.err; leg 1-bench $?

timeout -k 10 900 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/rp.err; rc=$?
if [ $rc -eq 0 ]; then
  F=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
  python tools/timed_region_stats.py $F 3 > $O/timed_region.txt; rc=$?
  cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv
  cp $(find $O/tr -name "b_marker*stats*.csv" | head -1) $O/marker_stats.csv 2>/dev/null
  python tools/gap_analysis.py $F 4 > $O/gaps.txt 2>&1 || true        # W + K = 4 factor() calls of the headline
fi
rm -rf $O/tr; leg 2-kernel-trace $rc

for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -c1-2 | tr A-Z a-z)
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$t -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr --no-config2 > $O/pmc_$t.json 2> $O/pmc_$t.err; rc=$?
  if [ $rc -eq 0 ]; then
    grep -E "Counter_Name|dgemm_tile_kernel<128, true, true>" $(find $O/pmc_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_all.csv
    python - <<PY
import json
n = int(json.loads([l for l in open('$O/pmc_$t.json') if l.startswith('{')][-1])['roofline']['launches_per_step']); c = 2 if '$t' == 'sq' else 1
l = open('$O/pmc_${t}_all.csv').read().splitlines()
open('$O/pmc_${t}_bench_n65536.csv', 'w').write('\n'.join(l[:1 + n * c]) + '\n')     # the factor() launches only (the validator's products follow)
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
done

# the round-2 crash case, once: the whole bench process (config 2 and both TRSM legs included) under --pmc.  Its outcome is RECORDED, not required.
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_whole -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_whole.json 2> $O/pmc_whole.err; rc=$?
echo "leg 6-whole-bench-under-pmc rc=$rc (recorded only; round 2: host SIGSEGV inside the profiler's dispatch interception)" | tee -a $O/legs.txt
tail -25 $O/pmc_whole.err > $O/pmc_whole_tail.err; rm -rf $O/pmc_whole $O/pmc_whole.err

ls -la $O; cat $O/legs.txt; cat $O/timed_region.txt; cut -c1-300 $O/bench.json
exit $fail
