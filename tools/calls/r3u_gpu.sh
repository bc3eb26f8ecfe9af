#!/bin/bash
# round 3, call u: the K-slice form of the 2-rank grid over the loopback transport once more, with a bounded wait and a heartbeat
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3u
mkdir -p $O
( while sleep 60; do echo "tick $(date +%T)"; done ) &
HB=$!
CAPITAL_KSLICE=1 CAPI_LOOPBACK_TIMEOUT_S=60 timeout -k 10 800 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "loopback2" > $O/multirank_kslice.log 2>&1; echo "multirank (K-slices) rc=$?" | tee -a $O/summary.txt
tail -30 $O/multirank_kslice.log | cut -c1-300
kill $HB
