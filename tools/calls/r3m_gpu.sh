#!/bin/bash
# round 3, call m: the headline step with every whole-round TRMM in tile pairs launched one resident round at a time, against the default
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
for i in 1 2; do
  python bench.py --steps 2 --no-cpu --no-qr > $O/default_$i.json 2> $O/default_$i.err; echo "default $i rc=$?" | tee -a $O/summary.txt
  CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1 python bench.py --steps 2 --no-cpu --no-qr > $O/pairrounds_$i.json 2> $O/pairrounds_$i.err; echo "pairs in rounds $i rc=$?" | tee -a $O/summary.txt
done
CAPI_TRMM_PAIR=2 python bench.py --steps 2 --no-cpu --no-qr > $O/pair2_1.json 2> $O/pair2_1.err; echo "pairs free-running rc=$?" | tee -a $O/summary.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), j["roofline"]["launches_per_step"],
          "config2", round(j["config2"]["ms_per_step"], 1), round(j["config2"]["trsm_mode"]["ms_per_step"], 1), "trsm65536", round(j["cholesky_trsm_mode"]["ms_per_step"], 1))
PY
