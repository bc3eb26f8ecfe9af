#!/bin/bash
# round 4, call l: eight ranks as threads of two processes on one GPU: the 2 x 2 x 2 grid, multi-path transfers at P = 8, layout 1, 3-D CholeskyQR2
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4l; rm -rf $O; mkdir -p $O/logs
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs CAPITAL_TEST_RANK_TIMEOUT_S=500 timeout -k 10 1100 python -m pytest tests/test_gpu_multirank.py -x -q -k "eight_ranks" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt
tail -30 $O/pytest.log; for f in $O/logs/*.log; do echo "== $f"; tail -n 4 $f; done
