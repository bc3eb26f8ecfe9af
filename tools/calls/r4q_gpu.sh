#!/bin/bash
# round 4, call q: lookahead off by default: whole GPU suite, then rounds ON (default) against one launch per product on one stream, alternating
export TMPDIR=/tmp
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -3 $O/pytest.log
one() {
  label=$1; shift
  env "$@" timeout -k 10 400 python bench.py --steps 3 --no-cpu --no-qr > $O/b.json 2> $O/b.err || echo "$label failed" | tee -a $O/legs.txt
  python - >> $O/ab.txt <<PY
import json
j = json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1]); r = j["roofline"]
print("$label: %.2f ms/step, group frac %.4f, launches/step %.0f, config2 %.2f ms (TRSM mode %.2f), TRSM mode n=65536 %.1f ms" % (j["ms_per_step"], r["frac"], r["launches_per_step"], j["config2"]["ms_per_step"], j["config2"]["trsm_mode"]["ms_per_step"], j["cholesky_trsm_mode"]["ms_per_step"]))
PY
}
for rep in 1 2; do one "rounds (default)" X=1; one "one launch per product" CAPITAL_NO_LAUNCH_ROUNDS=1; done
cat $O/ab.txt
