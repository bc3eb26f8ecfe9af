#!/bin/bash
# round 3, call c: T-stationary right-TRMM with even/odd row halves (no lane exchange before the stores); per-kernel times of CholeskyQR2
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py tests/test_golden.py -x -q -m gpu -k "panel32 or tall or cacqr or qr" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
for i in 1 2; do
  python tools/qr_ab.py 22 256 10 >> $O/qr.log 2>&1
  CAPITAL_NO_PANEL32=1 python tools/qr_ab.py 22 256 10 >> $O/qr.log 2>&1
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o qr -- python3 $OLDPWD/tools/qr_ab.py 22 256 10 > $OLDPWD/$O/prof.log 2>&1; cd $OLDPWD
find $O/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/qr_kernel_stats.csv
rm -rf $O/prof
grep -v amdgpu.ids $O/qr.log; head -8 $O/qr_kernel_stats.csv | cut -c1-200
