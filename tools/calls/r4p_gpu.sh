#!/bin/bash
# round 4, call p: is the single-GPU lookahead (bulk of the trailing updates on low-priority streams) still worth anything with launches in resident rounds?
# n = 65536 and n = 32768, alternating, one box; then the dependent-launch latency knobs of the HIP runtime on the blocked diagonal-block routine
export TMPDIR=/tmp
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
one() {
  label=$1; shift
  env "$@" timeout -k 10 300 python bench.py $BARGS --no-cpu --no-qr --no-config2 > $O/b.json 2> $O/b.err || echo "$label failed" | tee -a $O/legs.txt
  python - >> $O/lookahead.txt <<PY
import json
j = json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1]); r = j["roofline"]
print("$label $BARGS: %.2f ms/step, group frac %.4f (union %.1f ms), by event brackets %.2f TF/s" % (j["ms_per_step"], r["frac"], r["union_ms_per_step"], r["by_event_brackets"]["achieved"]))
PY
}
for rep in 1 2; do
  BARGS="--steps 3"; one "lookahead(default)" X=1; one "no-lookahead" CAPITAL_NO_LOOKAHEAD=1
  BARGS="--n 32768 --steps 6"; one "lookahead(default)" X=1; one "no-lookahead" CAPITAL_NO_LOOKAHEAD=1
done
cat $O/lookahead.txt
for v in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v" >> $O/pt.txt; HIP_FORCE_DEV_KERNARG=$v python tools/pt_bench.py >> $O/pt.txt 2>&1; done
cat $O/pt.txt | grep -v amdgpu.ids
