#!/bin/bash
# round 3, call 4b: one bulk stream (CAPITAL_LA_DEPTH=1) with and without launches in resident rounds, against the default (two bulk streams, no rounds)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
R="CAPI_ROUNDS=3 CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1"
for i in 1 2; do
  python bench.py --steps 2 --no-cpu --no-qr > $O/default_$i.json 2> $O/default_$i.err
  CAPITAL_LA_DEPTH=1 python bench.py --steps 2 --no-cpu --no-qr > $O/depth1_$i.json 2> $O/depth1_$i.err
  env $R CAPITAL_LA_DEPTH=1 python bench.py --steps 2 --no-cpu --no-qr > $O/depth1rounds_$i.json 2> $O/depth1rounds_$i.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_[12].json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), j["roofline"]["launches_per_step"], round(j["roofline"]["tile_kernel_share_of_step"], 3),
          "config2", round(j["config2"]["ms_per_step"], 1))
PY
