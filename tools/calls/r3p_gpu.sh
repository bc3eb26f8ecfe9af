#!/bin/bash
# round 3, call p: the headline step with EVERY large launch one resident round at a time (plain + triangular outputs, TRMMs in pairs): time,
# the roofline figure from HIP events, and 2 x FETCH_SIZE of the whole factor() against the default
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
R="CAPI_ROUNDS=3 CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1"
for i in 1 2; do
  python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/default_$i.json 2> $O/default_$i.err
  env $R python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/rounds_$i.json 2> $O/rounds_$i.err
done
env CAPI_ROUNDS=2 CAPI_TRMM_PAIR_ROUNDS=1 CAPI_TRMM_PAIR_ROUNDS_MIN=16384 python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/rounds_tri_only_1.json 2> $O/rounds_tri_only_1.err
for v in default rounds; do
  if [ $v = rounds ]; then export CAPI_ROUNDS=3 CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$v -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-qr --no-config2 > $O/pmc_$v.json 2> $O/pmc_$v.err
  python - <<PY >> $O/fetch.log
import csv, collections
f = "$(find $O/pmc_$v -name 'p_counter_collection.csv' | head -1)"
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the factor() call ends with its 5th packing kernel (serialize_kernel); what follows is the validator
ser = [i for i, r in enumerate(rows) if "serialize_kernel" in r["Kernel_Name"]]
rows = rows[:ser[4] + 1] if len(ser) >= 5 else rows
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    tot[k] += float(r["Counter_Value"]) * 1024 * 2 / 1e9; cnt[k] += 1
print("$v: 2 x FETCH_SIZE of one factor() at n = 65536:", round(sum(tot.values()), 1), "GB in", sum(cnt.values()), "launches")
for k, v in tot.most_common(6):
    print("     %9.1f GB  %6d  %s" % (v, cnt[k], k))
PY
  rm -rf $O/pmc_$v
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*_[12].json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], round(j["ms_per_step"], 1), round(j["value"], 2), "roofline", round(j["roofline"]["frac"], 4), j["roofline"]["launches_per_step"], round(j["roofline"]["tile_kernel_share_of_step"], 3))
PY
cat $O/fetch.log
