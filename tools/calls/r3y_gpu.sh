#!/bin/bash
# round 3, call y: the whole GPU suite, smoke, and the bench line at the round's final code
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
( while sleep 60; do echo "tick $(date +%T)"; done ) &
HB=$!
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
kill $HB
python - <<PY
import json
j = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print(round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), j["roofline"]["traffic"], j["roofline"]["traffic_source"])
for k in ("config2", "cacqr2", "cacqr2_config5", "cholesky_trsm_mode"):
    print("  ", k, {a: round(b, 3) for a, b in j[k].items() if a in ("tflops", "ms", "ms_per_step")})
PY
