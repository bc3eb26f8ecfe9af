#!/bin/bash
# round 4, call zi: legs 2 and 6 of tools/capture_profiles_r4.sh again at the round's last product commit (the leaf chain changed): kernel trace + stats of the bench command,
# timed-region statistics of the roofline symbols, idle analysis at n = 65536 and n = 32768
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4zi; rm -rf $O; mkdir -p $O
TILE='dgemm_tile_kernel<128, true, true>'
PAIR='dtrmm_pair_kernel<true, true>'
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/rp.err; rc=$?
if [ $rc -eq 0 ]; then
  F=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
  { python tools/timed_region_stats.py $F 3 1 "$TILE"; python tools/timed_region_stats.py $F 3 1 "$PAIR"; python tools/timed_region_stats.py $F 3 1 "$TILE|$PAIR"; } > $O/timed_region.txt; rc=$?
  cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python tools/gap_analysis.py $F 4 > $O/gaps_n65536.txt 2>&1 || true
fi
rm -rf $O/tr; echo "leg 2-kernel-trace rc=$rc" | tee -a $O/legs.txt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr2 -o b -- python bench.py --n 32768 --steps 2 --no-cpu --no-qr > $O/bench_n32768_under_rocprof.json 2> $O/rp2.err; rc=$?
if [ $rc -eq 0 ]; then python tools/gap_analysis.py $(find $O/tr2 -name "b_kernel_trace.csv" | head -1) 3 > $O/gaps_n32768.txt 2>&1; rc=$?; fi
rm -rf $O/tr2; echo "leg 6-gaps-n32768 rc=$rc" | tee -a $O/legs.txt
head -3 $O/gaps_n65536.txt | cut -c1-200; head -3 $O/gaps_n32768.txt | cut -c1-200; cat $O/timed_region.txt | tail -6 | cut -c1-250
git rev-parse --short HEAD 2>/dev/null > $O/commit.txt || true
