#!/bin/bash
# round 4, call n: eight ranks as threads, both transport modes, with deferred frees
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4n; rm -rf $O; mkdir -p $O/logs
CAPI_LOOPBACK_TIMEOUT_S=60 CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs CAPITAL_TEST_RANK_TIMEOUT_S=400 timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -k "eight_ranks" > $O/pytest.log 2>&1; echo "eight thread-ranks rc=$?" | tee -a $O/legs.txt
tail -5 $O/pytest.log
for f in $O/logs/*.log; do echo "== $f"; tail -n 14 $f | cut -c1-300; done
