#!/bin/bash
# round 3, call g: in-process A/B of the panel32 intermediate, on both builds of the T-stationary kernel; the workspace-OOM fallback test
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
python -m pytest tests/test_gpu_lapack.py -x -q -m gpu -k "geqrf or graph" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
for i in 1 2; do
  python tools/qr_ab2.py 22 20 >> $O/ab2.log 2>&1
  CAPITAL_HIP_LIB=$PWD/capital_amd/ab_swap_libcapital_hip.so python tools/qr_ab2.py 22 20 >> $O/ab2.log 2>&1
done
grep -v amdgpu.ids $O/ab2.log
