#!/bin/bash
# round 4, call v: where the GPU suite's eleven minutes go
export TMPDIR=/tmp
O=gpurun_out/r4v; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=30 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt
grep -A40 "slowest" $O/pytest.log | head -45; tail -3 $O/pytest.log
