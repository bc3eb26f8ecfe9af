#!/bin/bash
# round 3, call o: only the order-32768 TRMM in pairs + rounds inside the headline step; per-launch table of both forms; then call n's legs
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
for i in 1 2; do
  python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/default_$i.json 2> $O/default_$i.err
  CAPI_TRMM_PAIR_ROUNDS=1 CAPI_TRMM_PAIR_ROUNDS_MIN=32768 python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/pr32768_$i.json 2> $O/pr32768_$i.err
done
CAPI_PROF_DUMP=1 python bench.py --steps 1 --no-cpu --no-qr --no-config2 > $O/dump_default.json 2> $O/dump_default.err
CAPI_PROF_DUMP=1 CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1 python bench.py --steps 1 --no-cpu --no-qr --no-config2 > $O/dump_pairrounds.json 2> $O/dump_pairrounds.err
python - <<PY
import json, glob, re, collections
for f in sorted(glob.glob("$O/*_[12].json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), j["roofline"]["launches_per_step"])
for tag in ("default", "pairrounds"):
    agg = collections.OrderedDict()
    for l in open("$O/dump_%s.err" % tag):
        m = re.match(r"\[capi prof\]\s+\d+ (\w+) v(\d+) M=(\d+) N=(\d+) K=(\d+)\s+([\d.]+) ms", l)
        if m and int(m.group(5)) >= 8192:
            k = (m.group(1), "pair" if int(m.group(2)) >= 16 else "tile", int(m.group(2)) & 3, m.group(3), m.group(4), m.group(5))
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(m.group(6))
    print(tag)
    for k, (c, ms) in agg.items():
        print("   ", *k, "launches", c, "ms", round(ms, 2))
PY
bash tools/r3n_gpu.sh
