#!/bin/bash
# round 3, call x: the 2-rank loopback case first on a fresh box (after the oracle's thread cap), its K-slice form, bench N = 2 rehearsal
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3x
mkdir -p $O
( while sleep 60; do echo "tick $(date +%T)"; done ) &
HB=$!
timeout -k 10 500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/loopback2.log 2>&1; echo "loopback2 alone rc=$?" | tee -a $O/summary.txt
tail -2 $O/loopback2.log
CAPITAL_KSLICE=1 timeout -k 10 500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/loopback2_kslice.log 2>&1; echo "loopback2 alone, K-slices rc=$?" | tee -a $O/summary.txt
tail -2 $O/loopback2_kslice.log
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so timeout -k 10 600 python bench.py --gpus 2 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop2.json 2> $O/bench_loop2.err; echo "bench loopback N=2 rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
j = json.loads([l for l in open("$O/bench_loop2.json") if l.startswith("{")][-1])
print(j["n_gpus"], round(j["ms_per_step"], 1), j["config"]["residual"], j.get("cholesky_trsm_mode", {}).get("residual"))
for c in j["config"]["comm_forms"]:
    print("  ", c.get("form"), c.get("base_case_order"), round(c.get("ms_per_step", 0), 1), c.get("valid"), c.get("timed"), c.get("error"))
PY
kill $HB
