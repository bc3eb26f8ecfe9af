#!/bin/bash
# round 3, call 4c: the blocked diagonal-block routine with one block of lookahead (CAPI_BLOCKED_LA=1): parity tests, time per order, the steps
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
CAPI_BLOCKED_LA=1 timeout -k 10 600 python -m pytest tests/test_gpu_lapack.py tests/test_gpu_schedules.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests (CAPI_BLOCKED_LA=1) rc=$rc" | tee -a $O/summary.txt
tail -3 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
  python tools/pt_bench.py >> $O/pt.log 2>&1
  CAPI_BLOCKED_LA=1 python tools/pt_bench.py >> $O/pt.log 2>&1
done
for i in 1 2; do
  python bench.py --steps 2 --no-cpu --no-qr > $O/default_$i.json 2> $O/default_$i.err
  CAPI_BLOCKED_LA=1 python bench.py --steps 2 --no-cpu --no-qr > $O/la_$i.json 2> $O/la_$i.err
done
grep -v amdgpu $O/pt.log | sed 's/^\[auto\]/[pt]/' | cut -c1-330
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_[12].json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), "config2", round(j["config2"]["ms_per_step"], 1), round(j["config2"]["trsm_mode"]["ms_per_step"], 1),
          "trsm65536", round(j["cholesky_trsm_mode"]["ms_per_step"], 1), "res", j["config"]["residual"], j["config2"]["residual"])
PY
