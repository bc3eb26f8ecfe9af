#!/bin/bash
# round 4, call ze: config 4's grid (2 x 2 x 2) at n = 16384 on one card, eight thread-ranks over the asynchronous loopback
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4ze; rm -rf $O; mkdir -p $O/logs
make -C tests/rccl_loopback -s; make -C tests/thread_ranks -s
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs timeout -k 10 1000 python tools/rehearse_config4_grid.py 16384 $O/config4_grid.txt > $O/run.log 2>&1; echo "rehearsal rc=$?" | tee -a $O/legs.txt
cat $O/config4_grid.txt; grep -v amdgpu $O/run.log | tail -15 | cut -c1-600
