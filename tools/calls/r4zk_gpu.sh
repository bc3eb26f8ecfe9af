#!/bin/bash
# round 4, call zk: the eight-rank rehearsal at n = 24576 (12288 x 12288 per rank; rings of up to 2 GiB per channel)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4zk; rm -rf $O; mkdir -p $O/logs
CAPITAL_TEST_RANK_LOG_DIR=$PWD/$O/logs REHEARSE_ONLY=trsm_mode timeout -k 10 900 python tools/rehearse_config4_grid.py 24576 $O/config4_grid_n24576.txt > $O/run.log 2>&1; echo "rehearsal rc=$?" | tee -a $O/legs.txt
cat $O/config4_grid_n24576.txt; grep -v amdgpu $O/run.log | tail -12 | cut -c1-900; dmesg 2>/dev/null | tail -5
