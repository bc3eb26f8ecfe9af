#!/bin/bash
# round 4, call h: (1) the tall right-TRMM's band panels without their dead sub-tile columns (CAPI_NO_BAND_SKIP=1 = before), parity + timing;
# (2) resident rounds only from a minimum depth K (CAPI_ROUNDS_MIN_K) at n = 32768
export TMPDIR=/tmp
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_blas.py -x -q -k "wider_than_256 or tall_skinny or dtrmm" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -2 $O/pytest.log
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export CAPI_NO_BAND_SKIP=1; else unset CAPI_NO_BAND_SKIP; fi
  echo "== CAPI_NO_BAND_SKIP=$v" >> $O/ts_wide.txt
  timeout -k 10 200 python tools/ts_wide_bench.py 21 23 >> $O/ts_wide.txt 2>&1 || echo "ts_wide $v failed" | tee -a $O/legs.txt
done
unset CAPI_NO_BAND_SKIP
cat $O/ts_wide.txt
for k in none 0 8192 16384 32768 none 0 16384; do
  if [ $k = none ]; then export CAPITAL_NO_LAUNCH_ROUNDS=1; unset CAPI_ROUNDS_MIN_K; else unset CAPITAL_NO_LAUNCH_ROUNDS; export CAPI_ROUNDS_MIN_K=$k; fi
  timeout -k 10 200 python bench.py --n 32768 --steps 5 --no-cpu --no-qr > $O/b.json 2> $O/b.err || echo "bench min_k=$k failed" | tee -a $O/legs.txt
  python - >> $O/min_k.txt <<PY
import json
j = json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1]); r = j["roofline"]
print("rounds min K = $k: n=32768 %.2f ms/step, group frac %.4f, union %.1f ms, launches/step %.0f" % (j["ms_per_step"], r["frac"], r["union_ms_per_step"], r["launches_per_step"]))
PY
done
cat $O/min_k.txt
