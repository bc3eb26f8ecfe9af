#!/bin/bash
# round 3, call l: the big TRMMs in tile pairs, free-running against one launch per resident round: time and FETCH_SIZE
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1 python -m pytest tests/test_gpu_blas.py -x -q -m gpu -k "trmm or pair" > $O/tests.log 2>&1; rc=$?; echo "tests (pairs in rounds) rc=$rc" | tee -a $O/summary.txt
tail -2 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
run() {  # $1 = tag, rest = env assignments
  tag=$1; shift
  env "$@" python tools/trmm_fetch.py 16384 32768 >> $O/time.log 2>&1
}
run default
run pair2 CAPI_TRMM_PAIR=2
run pair2rounds CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1
run default
run pair2rounds CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1
for v in default pair2 pair2rounds; do
  case $v in
    default) export -n CAPI_TRMM_PAIR CAPI_TRMM_PAIR_ROUNDS; unset CAPI_TRMM_PAIR CAPI_TRMM_PAIR_ROUNDS;;
    pair2) export CAPI_TRMM_PAIR=2; unset CAPI_TRMM_PAIR_ROUNDS;;
    pair2rounds) export CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1;;
  esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 tools/trmm_fetch.py 16384 32768 > $O/pmc_$v.log 2>&1
  python - <<PY >> $O/fetch.log
import csv, collections
f = "$(find $O/pmc_$v -name 'p_counter_collection.csv' | head -1)"
tot = collections.Counter(); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"]
    if "trmm_pair" in k or "dgemm_tile" in k:
        key = ("pair" if "pair" in k else "tile") 
        tot[key] += float(r["Counter_Value"]) * 1024 * 2 / 1e9; cnt[key] += 1
print("$v", {k: (round(v, 1), cnt[k]) for k, v in tot.items()}, "GB (launches); 4 calls each of orders 16384 and 32768")
PY
  rm -rf $O/pmc_$v
done
grep -v amdgpu.ids $O/time.log; cat $O/fetch.log
