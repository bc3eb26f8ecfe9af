#!/bin/bash
# round 3, call w: the 2-rank loopback case as the FIRST thing on a fresh box, every rank's output kept (why it stalls there)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3w
mkdir -p $O
( while sleep 60; do echo "tick $(date +%T)"; tail -q -n 1 $O/rank*.log 2>/dev/null | cut -c1-160; done ) &
HB=$!
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O CAPITAL_TEST_RANK_TIMEOUT_S=240 CAPITAL_TEST_GLOO_TIMEOUT_S=120 timeout -k 10 560 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/pytest.log 2>&1; echo "loopback2 alone rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | cut -c1-300
for f in $O/rank*.log; do echo "== $f"; tail -12 $f | cut -c1-300; done
kill $HB
