#!/bin/bash
# round 4, call g: (1) the bench line with the roofline taken over both 128-tile TN symbols; (2) the CU-contention experiment
# (tools/overlap/overlap_probe.hip); (3) the dispatch-count probe of the rocprofv3 --pmc SIGSEGV; (4) leaf phase trace, pt_bench
export TMPDIR=/tmp
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
timeout -k 10 500 python bench.py --steps 3 --no-cpu --no-qr > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/legs.txt
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r4g/bench.json") if l.startswith("{")][-1]); r = j["roofline"]
print("ms/step", round(j["ms_per_step"], 1), "frac", round(r["frac"], 4), "union", round(r["union_ms_per_step"], 1), [(s["symbol"][:20], round(s["launches_per_step"]), round(s["avg_launch_ms"], 3), round(s["rate_over_own_union"] or 0, 2)) for s in r["symbols"]], "cfg2", j["config2"]["ms_per_step"], j["config2"]["roofline_kernel_tflops"])
PY
make -C tools/overlap -s
for cfg in "16384 4 16 60 0" "16384 8 16 60 0" "16384 4 16 60 8" "16384 4 16 60 16" "16384 4 16 60 32" "16384 8 16 60 16" "16384 4 8 60 16" "16384 4 32 0 0" "16384 4 32 0 16"; do
  timeout -k 10 120 tools/overlap/overlap_probe $cfg >> $O/overlap.txt 2>> $O/overlap.err || echo "overlap $cfg failed rc=$?" | tee -a $O/legs.txt
done
for cfg in "16384 4 16 60 0" "16384 8 16 60 0"; do
  CAPI_COMM_PRIO_NORMAL=1 timeout -k 10 120 tools/overlap/overlap_probe $cfg >> $O/overlap.txt 2>> $O/overlap.err || echo "overlap prio-normal $cfg failed" | tee -a $O/legs.txt
done
cat $O/overlap.txt
# the profiler alone: 120000 trivial dispatches under --pmc (no product code in the process)
make -C tools/pmc_probe -s 2>/dev/null || /opt/rocm/bin/hipcc -O1 --offload-arch=gfx950 tools/pmc_probe/dispatch_count_probe.hip -o tools/pmc_probe/dispatch_count_probe
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_count -o p -- tools/pmc_probe/dispatch_count_probe 120000 > $O/pmc_count.out 2> $O/pmc_count.err; echo "pmc dispatch-count probe rc=$?" | tee -a $O/legs.txt
tail -3 $O/pmc_count.out; tail -5 $O/pmc_count.err; find $O/pmc_count -type f | head; rm -rf $O/pmc_count
python tools/leaf_trace.py > $O/leaf_trace.txt 2>&1; tail -3 $O/leaf_trace.txt
python tools/pt_bench.py > $O/pt_bench.txt 2>&1; cat $O/pt_bench.txt
