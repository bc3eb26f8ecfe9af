#!/bin/bash
# round 3, call r: grids with launches in resident rounds (loopback 2 / 4 ranks), blas tests, bench N = 4 rehearsal
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/multirank.log 2>&1; echo "multirank rc=$?" | tee -a $O/summary.txt
tail -3 $O/multirank.log
python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py -x -q -m gpu > $O/blas.log 2>&1; echo "blas+schedules rc=$?" | tee -a $O/summary.txt
tail -2 $O/blas.log
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPITAL_MULTIPATH_MIN=4096 timeout -k 10 900 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop4.json 2> $O/bench_loop4.err; echo "bench loopback N=4 rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
j = json.loads([l for l in open("$O/bench_loop4.json") if l.startswith("{")][-1])
print(j["n_gpus"], round(j["ms_per_step"], 1), j["config"]["residual"], j["config"]["launches_in_resident_rounds"], j.get("cholesky_trsm_mode", {}).get("residual"))
for c in j["config"]["comm_forms"]:
    print("  ", c.get("form"), c.get("base_case_order"), round(c.get("ms_per_step", 0), 1), c.get("valid"), c.get("timed"))
PY
