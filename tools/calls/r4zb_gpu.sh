#!/bin/bash
# round 4, call zb: the whole multirank file with labelled rings (allocation = capacity + 4096, label verified at import), default settings
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4zb; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "multirank rc=$?" | tee -a $O/legs.txt
tail -25 $O/pytest.log | cut -c1-400
