#!/bin/bash
# round 3, last call: tall-skinny + RCCL single-rank tests at the final source (the T-stationary kernel was edited once more), CholeskyQR2 timing
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py tests/test_gpu_rccl.py tests/test_golden.py tests/test_gpu_lapack.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
python tools/qr_ab2.py 22 15 > $O/ab2.log 2>&1; grep cacqr2 $O/ab2.log
