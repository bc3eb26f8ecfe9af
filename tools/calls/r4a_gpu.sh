#!/bin/bash
# round 4, call a: the asynchronous loopback transport -- parity cases on 2 and 4 ranks, then the negative test
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r4a_logs
export CAPITAL_TEST_RANK_LOG_DIR=$PWD/gpurun_out/r4a_logs CAPITAL_TEST_RANK_TIMEOUT_S=400
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -k "async" > gpurun_out/r4a_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4a_pytest.log
tail -40 gpurun_out/r4a_pytest.log
