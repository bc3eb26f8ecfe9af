#!/bin/bash
# round 4, call o: the schedule's knobs at n = 32768 with launches in resident rounds (the new default): lookahead threshold / depth, base-case order
export TMPDIR=/tmp
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --n 32768 --steps 5 --no-cpu --no-qr $BARGS > $O/b.json 2> $O/b.err || echo "$label failed" | tee -a $O/legs.txt
  python - >> $O/sweep.txt <<PY
import json
j = json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
print("$label: %.2f ms/step, base case %s, group frac %.4f" % (j["ms_per_step"], j["config"]["base_case_order"], j["roofline"]["frac"]))
PY
}
BARGS=""
run "default" X=1
run "LA_MIN=1024" CAPITAL_LOOKAHEAD_MIN=1024
run "LA_MIN=4096" CAPITAL_LOOKAHEAD_MIN=4096
run "LA_DEPTH=1" CAPITAL_LA_DEPTH=1
run "no lookahead" CAPITAL_NO_LOOKAHEAD=1
BARGS="--bc -4"; run "base case 2048" X=1
BARGS="--bc -3"; run "base case 4096" X=1
BARGS="--bc -6"; run "base case 512" X=1
BARGS=""
run "default again" X=1
cat $O/sweep.txt
