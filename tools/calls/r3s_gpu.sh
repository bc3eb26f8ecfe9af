#!/bin/bash
# round 3, call s: config 5's tall products (n = 1024) with the block launches one resident round at a time: time and FETCH_SIZE
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3s
mkdir -p $O
for v in 0 7 0 7; do CAPI_ROUNDS=$v python tools/ts_wide_bench.py 21 22 --n 1024 >> $O/time.log 2>&1; done
for v in 0 7; do
  CAPI_ROUNDS=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 tools/ts_wide_bench.py 21 --n 1024 > $O/pmc_$v.log 2>&1
  python - <<PY >> $O/fetch.log
import csv, collections
f = "$(find $O/pmc_$v -name 'p_counter_collection.csv' | head -1)"
tot = collections.Counter(); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    tot[k] += float(r["Counter_Value"]) * 1024 * 2 / 1e9; cnt[k] += 1
print("CAPI_ROUNDS=$v (whole process: 2 checks + 4 timed calls of each product at m = 2^21):", {k: (round(v, 1), cnt[k]) for k, v in tot.most_common(5)})
PY
  rm -rf $O/pmc_$v
done
grep -v "check\|amdgpu" $O/time.log; cat $O/fetch.log
