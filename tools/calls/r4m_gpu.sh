#!/bin/bash
# round 4, call m: why do eight thread-ranks stall in the asynchronous loopback mode?  (a) four ranks as 2 processes x 2 threads on 2 x 2 x 1;
# (b) eight ranks again with the watchdog naming every channel that has queued work
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4m; rm -rf $O; mkdir -p $O/logs4 $O/logs8
cat > /tmp/four.py <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tests")
import test_gpu_multirank as T
import oracle as O
O.build()
cases = [{"tag": "ch_p0", "kind": "cholinv", "n": 4096, "c": 1, "bc": -3, "ci": 1, "serialize": True, "policy": 0},
         {"tag": "ch_chunks", "kind": "cholinv", "n": 4096, "c": 1, "bc": -3, "ci": 1, "serialize": True, "policy": 0, "chunks": 3}]
with tempfile.TemporaryDirectory() as d:
    T._launch_thread_ranks(2, 2, {"dir": d, "cases": cases}, "async", timeout=300)
    T._check_cases(O, d, cases, 4, 1)
print("four thread-ranks (2 x 2 x 1) in async mode: ok")
PY
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs4 timeout -k 10 400 python /tmp/four.py > $O/four.log 2>&1; echo "four thread-ranks rc=$?" | tee -a $O/legs.txt; tail -5 $O/four.log
for f in $O/logs4/*.log; do echo "== $f"; tail -n 12 $f | cut -c1-400; done
CAPI_LOOPBACK_TIMEOUT_S=45 CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs8 CAPITAL_TEST_RANK_TIMEOUT_S=300 timeout -k 10 500 python -m pytest tests/test_gpu_multirank.py -x -q -k "eight_ranks and async" > $O/pytest8.log 2>&1; echo "eight thread-ranks async rc=$?" | tee -a $O/legs.txt
for f in $O/logs8/*.log; do echo "== $f"; tail -n 30 $f | cut -c1-400; done
