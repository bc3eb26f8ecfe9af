#!/bin/bash
# round 3, call 4e: bench.py N = 4 and N = 2 rehearsals over the loopback transport at the final bench.py
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
for N in 4 2; do
  CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPITAL_MULTIPATH_MIN=4096 timeout -k 10 600 python bench.py --gpus $N --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop$N.json 2> $O/bench_loop$N.err; echo "bench loopback N=$N rc=$?" | tee -a $O/summary.txt
  python - <<PY
import json
j = json.loads([l for l in open("$O/bench_loop$N.json") if l.startswith("{")][-1])
print(j["n_gpus"], round(j["ms_per_step"], 1), j["config"]["residual"], j["config"]["summa_chunks"], j["config"]["multipath_pair_transfers"], j["config"]["base_case_order"], j.get("cholesky_trsm_mode", {}).get("residual"), j["cacqr2_config5"]["residual"])
for c in j["config"]["comm_forms"]:
    print("  ", c.get("form"), c.get("base_case_order"), round(c.get("ms_per_step", 0), 1), c.get("valid"), c.get("timed"), c.get("note"), c.get("error"))
PY
done
