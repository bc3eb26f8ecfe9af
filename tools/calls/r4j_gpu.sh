#!/bin/bash
# round 4, call j: the product's own copy kernel instead of the runtime's blit kernels: (1) whole GPU suite; (2) bench line, A/B against
# CAPI_RUNTIME_COPY=1 on the same box; (3) the whole bench process under rocprofv3 --pmc ONCE: the test of that change (crash record: r4_pmc_whole_bench_crash.txt)
export TMPDIR=/tmp
O=gpurun_out/r4j; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -3 $O/pytest.log
timeout -k 10 600 python bench.py --steps 3 --no-cpu > $O/bench_own_copy.json 2> $O/bench_own_copy.err; echo "bench own copy rc=$?" | tee -a $O/legs.txt
CAPI_RUNTIME_COPY=1 timeout -k 10 600 python bench.py --steps 3 --no-cpu > $O/bench_runtime_copy.json 2> $O/bench_runtime_copy.err; echo "bench runtime copy rc=$?" | tee -a $O/legs.txt
python - <<'PY'
import json
for f in ("bench_own_copy", "bench_runtime_copy"):
    j = json.loads([l for l in open(f"gpurun_out/r4j/{f}.json") if l.startswith("{")][-1]); r = j["roofline"]
    print(f, "ms/step", round(j["ms_per_step"], 1), "frac", round(r["frac"], 4), "cfg2", round(j["config2"]["ms_per_step"], 2), "cfg2 trsm", round(j["config2"]["trsm_mode"]["ms_per_step"], 2),
          "trsm", round(j["cholesky_trsm_mode"]["ms_per_step"], 1), "qr", round(j["cacqr2"]["ms"], 3), "qr5", round(j["cacqr2_config5"]["ms"], 1))
PY
CAPITAL_BENCH_DUMP_MAPS=$O/pmc_whole_maps.txt timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_whole -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_whole.json 2> $O/pmc_whole.err; rc=$?
echo "whole bench under --pmc (own copy kernel) rc=$rc" | tee -a $O/legs.txt
{ echo "# files the profiler left behind:"; find $O/pmc_whole -type f -exec wc -l {} + 2>/dev/null; } > $O/pmc_whole_files.txt
if [ $rc -eq 0 ]; then python - <<PY
import csv, collections
f = "$(find $O/pmc_whole -name 'p_counter_collection.csv' | head -1)"
cnt = collections.Counter(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:] for r in csv.DictReader(open(f)))
print("dispatches profiled:", sum(cnt.values())); [print("  ", v, k) for k, v in cnt.most_common(12)]
PY
fi > $O/pmc_whole_dispatches.txt
rm -rf $O/pmc_whole
grep -E "^bench.py: leg|^\*\*\*|^PC:" $O/pmc_whole.err | head; cat $O/pmc_whole_dispatches.txt; cat $O/legs.txt
