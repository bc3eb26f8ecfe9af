#!/bin/bash
# round 3, call v: which case of the 2-rank loopback run stalls with CAPITAL_KSLICE=1 (rank-level progress, 200 s per launch)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
( while sleep 60; do echo "tick $(date +%T)"; done ) &
HB=$!
CAPITAL_TEST_RANK_TIMEOUT_S=200 CAPI_LOOPBACK_TIMEOUT_S=60 timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/default.log 2>&1; echo "loopback2 alone, default rc=$?" | tee -a $O/summary.txt
tail -5 $O/default.log | cut -c1-300
CAPITAL_KSLICE=1 CAPITAL_TEST_RANK_TIMEOUT_S=200 CAPI_LOOPBACK_TIMEOUT_S=60 timeout -k 10 400 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/kslice.log 2>&1; echo "loopback2 alone, K-slices rc=$?" | tee -a $O/summary.txt
grep -n "case\|Error\|error\|rank" $O/kslice.log | tail -30 | cut -c1-300
kill $HB
