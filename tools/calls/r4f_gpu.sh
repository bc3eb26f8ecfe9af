#!/bin/bash
# round 3, call 4f: where the steps go -- kernel trace of one bench process, gap analysis of the headline step and of the last config-2 step
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4f
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-qr > $O/bench.json 2> $O/rp.err; echo "trace rc=$?" | tee -a $O/summary.txt
F=$(find $O/tr -name "b_kernel_trace.csv" | head -1)
python tools/gap_analysis.py $F 2 > $O/gaps_n65536.txt 2>&1; echo "gaps 65536 rc=$?" | tee -a $O/summary.txt
python tools/gap_analysis.py $F 6 > $O/gaps_n32768.txt 2>&1; echo "gaps 32768 rc=$?" | tee -a $O/summary.txt
rm -rf $O/tr
head -12 $O/gaps_n65536.txt | cut -c1-220; head -12 $O/gaps_n32768.txt | cut -c1-220
