#!/bin/bash
# round 3, call k: the T-stationary TRMM with the next tile's LDS stores inside the MFMA loop (build A/B), per kernel and end to end
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
CAPITAL_HIP_LIB=$PWD/capital_amd/ab_stage_libcapital_hip.so python -m pytest tests/test_gpu_blas.py -x -q -m gpu -k "panel32 or tall" > $O/tests.log 2>&1; rc=$?; echo "tests (stage-in-loop build) rc=$rc" | tee -a $O/summary.txt
tail -2 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2 3; do
  CAPITAL_HIP_LIB=$PWD/capital_amd/ab_stage_libcapital_hip.so python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1 && \
  python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1
done
CAPITAL_HIP_LIB=$PWD/capital_amd/ab_stage_libcapital_hip.so python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1
python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1
grep -v amdgpu.ids $O/ab.log | grep trmm; grep -v amdgpu.ids $O/ab2.log
