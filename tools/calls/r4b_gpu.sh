#!/bin/bash
# round 4, call b: GPU suite (without the multirank file, run in call a), then the bench line with launches in resident rounds (the new
# default) against one launch per product, then the interval instrument against a rocprofv3 kernel trace of the same command
export TMPDIR=/tmp
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_multirank.py > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt
tail -3 $O/pytest.log
timeout -k 10 600 python bench.py --steps 3 --no-cpu > $O/bench_rounds.json 2> $O/bench_rounds.err; echo "bench rounds rc=$?" | tee -a $O/legs.txt
CAPITAL_NO_LAUNCH_ROUNDS=1 timeout -k 10 400 python bench.py --steps 3 --no-cpu --no-qr > $O/bench_norounds.json 2> $O/bench_norounds.err; echo "bench norounds rc=$?" | tee -a $O/legs.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 2 --no-cpu --no-qr --no-config2 > $O/bench_under_rocprof.json 2> $O/rp.err; rc=$?
echo "rocprof rc=$rc" | tee -a $O/legs.txt
if [ $rc -eq 0 ]; then
  F=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
  python tools/timed_region_stats.py $F 2 > $O/timed_region.txt
  cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python tools/gap_analysis.py $F 3 > $O/gaps.txt 2>&1 || true
fi
rm -rf $O/tr
python - <<'PY'
import json
for f in ("bench_rounds", "bench_norounds", "bench_under_rocprof"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r4b/{f}.json") if l.startswith("{")][-1])
        r = j["roofline"]
        print(f, "ms/step", round(j["ms_per_step"], 1), "TF", round(j["value"], 2), "frac", round(r["frac"], 4), "union/step", round(r["union_ms_per_step"], 1), "sum/step", round(r["sum_ms_per_step"], 1),
              "launches/step", r["launches_per_step"], "events:", round(r["by_event_brackets"]["achieved"], 2), "cfg2", j.get("config2", {}).get("ms_per_step"), "qr", j.get("cacqr2", {}).get("ms"), "qr5", j.get("cacqr2_config5", {}).get("tflops"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/timed_region.txt
