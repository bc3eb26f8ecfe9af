#!/bin/bash
# round 3, call z: the T-stationary TRMM with staging, prefetch (two tiles ahead) and stores BEFORE the barrier (build A/B), per kernel and end to end
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3z
mkdir -p $O
CAPITAL_HIP_LIB=$PWD/capital_amd/ab_wbb_libcapital_hip.so python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py -x -q -m gpu -k "panel32 or tall or cacqr or qr" > $O/tests.log 2>&1; rc=$?; echo "tests (work-before-barrier build) rc=$rc" | tee -a $O/summary.txt
tail -2 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2 3; do
  CAPITAL_HIP_LIB=$PWD/capital_amd/ab_wbb_libcapital_hip.so python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1 && \
  python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1
done
for i in 1 2; do
CAPITAL_HIP_LIB=$PWD/capital_amd/ab_wbb_libcapital_hip.so python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1
python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1
done
grep -v amdgpu.ids $O/ab.log | grep trmm; grep -v amdgpu.ids $O/ab2.log | grep cacqr2
