#!/bin/bash
# round 3, call e: full GPU suite at the new kernels / SUMMA (multi-path + pipelines through the loopback transport)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/multirank.log 2>&1; echo "multirank rc=$?" | tee -a $O/summary.txt
tail -5 $O/multirank.log
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_multirank.py > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/gpu_tests.log
