#!/bin/bash
# round 3, call h: LDS-DMA form of the T-stationary right-TRMM: parity tests, then per-kernel and end-to-end A/B
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
CAPI_TS32_DMA=1 timeout -k 10 600 python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py tests/test_golden.py -x -q -m gpu -k "panel32 or tall or cacqr or qr" > $O/tests_dma.log 2>&1; rc=$?; echo "tests (DMA form) rc=$rc" | tee -a $O/summary.txt
tail -3 $O/tests_dma.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
  CAPI_TS32_DMA=1 python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1 && \
  python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1 && \
  CAPI_TS32_DMA=1 python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1 && \
  python tools/qr_ab2.py 22 15 >> $O/ab2.log 2>&1
done
grep -v amdgpu.ids $O/ab.log | grep trmm; grep -v amdgpu.ids $O/ab2.log
