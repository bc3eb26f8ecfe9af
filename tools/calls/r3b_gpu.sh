#!/bin/bash
# round 3, call b: panel32 intermediate of CholeskyQR2 (n = 256) and the diagonal blocks of the tall right-TRMM (n = 1024) by the T-stationary kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
python -m pytest tests/test_gpu_blas.py -x -q -m gpu -k "panel32 or tall" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
for i in 1 2; do
  python tools/qr_ab.py 22 256 10 >> $O/qr.log 2>&1
  CAPITAL_NO_PANEL32=1 python tools/qr_ab.py 22 256 10 >> $O/qr.log 2>&1
done
python tools/qr_ab.py 21 1024 4 >> $O/qr.log 2>&1
CAPI_TALL_DIAG_TS=0 python tools/qr_ab.py 21 1024 4 >> $O/qr.log 2>&1
python tools/ts_wide_bench.py 21 --n 1024 >> $O/wide.log 2>&1
CAPI_TALL_DIAG_TS=0 python tools/ts_wide_bench.py 21 --n 1024 >> $O/wide.log 2>&1
python tools/ts_wide_bench.py 21 --n 512 >> $O/wide.log 2>&1
CAPI_TALL_DIAG_TS=0 python tools/ts_wide_bench.py 21 --n 512 >> $O/wide.log 2>&1
cat $O/qr.log; grep -v check $O/wide.log
