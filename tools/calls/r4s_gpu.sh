#!/bin/bash
# round 4, call s: after the ring flow-control fix: every multirank case (one rank per process and ranks as threads, both transport modes, negative test)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4s; rm -rf $O; mkdir -p $O/logs
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs CAPITAL_TEST_RANK_TIMEOUT_S=500 timeout -k 10 1100 python -m pytest tests/test_gpu_multirank.py -x -q > $O/pytest.log 2>&1; echo "multirank rc=$?" | tee -a $O/legs.txt
tail -4 $O/pytest.log
