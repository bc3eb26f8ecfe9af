#!/bin/bash
# round 4, call k: (1) whole GPU suite; (2) the diagnosis of the --pmc SIGSEGV put to the test: the fault address is the END of a 1 MiB queue ring
# (16384 AQL packets), so with a ring that never wraps inside the process (ROC_AQL_QUEUE_SIZE=131072) the whole bench must pass under --pmc;
# (3) per-launch table of the n = 32768 step; (4) the asynchronous loopback cases under stress (receive-side copies delayed, 2 copy workgroups)
export TMPDIR=/tmp
O=gpurun_out/r4k; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -3 $O/pytest.log
ROC_AQL_QUEUE_SIZE=131072 CAPITAL_BENCH_DUMP_MAPS=$O/pmc_whole_maps.txt timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_whole -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_whole.json 2> $O/pmc_whole.err; rc=$?
echo "whole bench under --pmc with ROC_AQL_QUEUE_SIZE=131072 rc=$rc" | tee -a $O/legs.txt
if [ $rc -eq 0 ]; then python - <<PY
import csv, collections
f = "$(find $O/pmc_whole -name 'p_counter_collection.csv' | head -1)"
cnt = collections.Counter(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:] for r in csv.DictReader(open(f)))
print("dispatches profiled:", sum(cnt.values())); [print("  ", v, k) for k, v in cnt.most_common(14)]
PY
fi > $O/pmc_whole_dispatches.txt
rm -rf $O/pmc_whole; grep -c "rw-s" $O/pmc_whole_maps.txt > $O/pmc_whole_maps_rws_count.txt; grep -E " 1024 |00100000" /dev/null; awk '{split($1,a,"-"); if (strtonum("0x" a[2]) - strtonum("0x" a[1]) >= 1048576 && $2 == "rw-s") print}' $O/pmc_whole_maps.txt 2>/dev/null | head -20 > $O/pmc_whole_maps_rings.txt; rm -f $O/pmc_whole_maps.txt
grep -E "^bench.py: leg|^\*\*\*|^PC:" $O/pmc_whole.err | head; cat $O/pmc_whole_dispatches.txt
CAPI_PROF_DUMP=1 timeout -k 10 300 python bench.py --n 32768 --steps 1 --no-cpu --no-qr > $O/dump_n32768.json 2> $O/dump_n32768.err; echo "prof dump rc=$?" | tee -a $O/legs.txt
grep "capi prof iv" $O/dump_n32768.err | awk '{print $0}' > $O/prof_iv_n32768.txt; wc -l $O/prof_iv_n32768.txt
mkdir -p $O/logs
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs CAPITAL_TEST_RANK_TIMEOUT_S=500 CAPI_LOOPBACK_DELAY_US=800 CAPI_LOOPBACK_COPY_WGS=2 timeout -k 10 1000 python -m pytest tests/test_gpu_multirank.py -x -q -k "loopback2_async or loopback4_async" > $O/pytest_async_stress.log 2>&1; echo "async stress rc=$?" | tee -a $O/legs.txt; tail -3 $O/pytest_async_stress.log
cat $O/legs.txt
