#!/bin/bash
# round 4, call w: why do the thread-rank tests take 5 x longer inside the whole suite than alone?  the n = 65536 test first, then the two tests, rank logs kept
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4w; rm -rf $O; mkdir -p $O/logs
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -k "n65536 or eight_ranks" --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt
grep -A8 "slowest" $O/pytest.log; tail -2 $O/pytest.log
grep -h "starts at\|ok at" $O/logs/threads_proc0_of4_async.log $O/logs/threads_proc0_of4_host.log | head -40
