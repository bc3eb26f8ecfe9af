#!/bin/bash
# round 4, call zc: why do the eight-rank runs take 90-106 s inside the whole suite and 11-30 s in their own file?  (a) after the full-size tests only, (b) after everything before them
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4zc; rm -rf $O; mkdir -p $O/a $O/b
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/a timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -x -q -k "fullsize or eight_ranks" --durations=8 > $O/a.log 2>&1; echo "a rc=$?" | tee -a $O/legs.txt
tail -12 $O/a.log | cut -c1-200; grep -h "starts at\|ok at" $O/a/threads_proc0_of4_*.log
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/b timeout -k 10 500 python -m pytest tests/test_gpu_lapack.py tests/test_gpu_movement.py tests/test_gpu_multirank.py -m gpu -x -q -k "lapack or movement or eight_ranks" --durations=4 > $O/b.log 2>&1; echo "b rc=$?" | tee -a $O/legs.txt
tail -8 $O/b.log | cut -c1-200; grep -h "starts at\|ok at" $O/b/threads_proc0_of4_*.log
