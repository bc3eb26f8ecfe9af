#!/bin/bash
# round 4, call i: (1) ONE recorded attempt of the whole bench process under rocprofv3 --pmc with the process's memory map kept (leg 7 of
# tools/capture_profiles_r4.sh, taken early so that its record can be acted on); (2) bench.py N = 2 / 4 rehearsals over the loopback transport
# in its host-staged and asynchronous modes (per-probe timers, budget, launches in rounds on grids)
export TMPDIR=/tmp
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
CAPITAL_BENCH_DUMP_MAPS=$O/pmc_whole_maps.txt timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_whole -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_whole.json 2> $O/pmc_whole.err; rc=$?
echo "whole bench under --pmc rc=$rc" | tee -a $O/legs.txt
{ echo "# files the profiler left behind:"; find $O/pmc_whole -type f -exec wc -l {} + 2>/dev/null; } > $O/pmc_whole_files.txt
rm -rf $O/pmc_whole
{ head -1 $O/pmc_whole_maps.txt; grep "r-xp" $O/pmc_whole_maps.txt | awk '{print $1, $3, $6}'; } > $O/pmc_whole_maps_modules.txt; rm -f $O/pmc_whole_maps.txt
grep -n "leg \|^\*\*\*\|SIG\|@ " $O/pmc_whole.err | head -60
L=$PWD/tests/rccl_loopback/librccl_loopback.so
for mode in host async; do
  CAPI_LOOPBACK_MODE=$mode CAPI_LOOPBACK_TIMEOUT_S=90 CAPI_RCCL_LIB=$L timeout -k 10 600 python bench.py --gpus 2 --one-device --n 8192 --steps 2 --no-cpu --no-qr > $O/bench_loop2_$mode.json 2> $O/bench_loop2_$mode.err; echo "bench loopback N=2 $mode rc=$?" | tee -a $O/legs.txt
  CAPI_LOOPBACK_MODE=$mode CAPI_LOOPBACK_TIMEOUT_S=90 CAPI_RCCL_LIB=$L CAPITAL_MULTIPATH_MIN=4096 timeout -k 10 600 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop4_$mode.json 2> $O/bench_loop4_$mode.err; echo "bench loopback N=4 $mode rc=$?" | tee -a $O/legs.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4i/bench_loop*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "ms/step", round(j["ms_per_step"], 2), "residual", j["config"]["residual"], "forms", [(c.get("form"), round(c.get("ms_per_step", 0), 1), c.get("valid", c.get("skipped", c.get("error", "")))) for c in j["config"]["comm_forms"]], "skipped", j.get("skipped_for_budget"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
cat $O/legs.txt
