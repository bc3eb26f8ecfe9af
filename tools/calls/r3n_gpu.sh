#!/bin/bash
# round 3, call n: triangular outputs (and the lookahead's rectangle) one resident round per launch: time and FETCH_SIZE
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
CAPI_ROUNDS=3 python -m pytest tests/test_gpu_blas.py -x -q -m gpu -k "syrk or gemmt or banded or dgemm" > $O/tests.log 2>&1; rc=$?; echo "tests (CAPI_ROUNDS=3) rc=$rc" | tee -a $O/summary.txt
tail -2 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 0 3 0 3; do CAPI_ROUNDS=$v python tools/syrk_fetch.py >> $O/time.log 2>&1; done
for v in 0 3; do
  CAPI_ROUNDS=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 tools/syrk_fetch.py > $O/pmc_$v.log 2>&1
  python - <<PY >> $O/fetch.log
import csv
f = "$(find $O/pmc_$v -name 'p_counter_collection.csv' | head -1)"
tot = 0.0; n = 0
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "dgemm_tile_kernel<128" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]) * 1024 * 2 / 1e9; n += 1
print("CAPI_ROUNDS=$v: 2 x FETCH_SIZE of the 128-tile launches", round(tot, 1), "GB in", n, "launches (4 calls of each of the four products)")
PY
  rm -rf $O/pmc_$v
done
grep -v amdgpu.ids $O/time.log; cat $O/fetch.log
