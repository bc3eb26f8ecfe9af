#!/bin/bash
# round 3, call d: per-kernel A/B of the T-stationary TRMM (lane-swap build vs even/odd-halves build), kernel stats of CholeskyQR2
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
for i in 1 2 3; do
  python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1
  CAPITAL_HIP_LIB=$PWD/capital_amd/ab_swap_libcapital_hip.so python tools/ts_ab.py 22 20 >> $O/ab.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o qr -- python3 tools/qr_ab.py 22 256 10 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $O/qr_kernel_stats.csv
rm -rf $O/prof
grep -v amdgpu.ids $O/ab.log; head -8 $O/qr_kernel_stats.csv | cut -c1-220
