#!/bin/bash
# round 4, call zj: two accumulator chains in dgemm_small_kernel (a wave owns one 16 x 16 tile: K / 4 dependent MFMAs) against the single chain (build/base): parity, per-call times, order 512..8192
export TMPDIR=/tmp
O=gpurun_out/r4zj; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_blas.py tests/test_gpu_lapack.py tests/test_golden.py -m gpu -x -q > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/legs.txt; tail -2 $O/parity.log | cut -c1-200
grep -q "parity rc=0" $O/legs.txt || exit 1
B=$PWD/build/base
for v in base new base new; do
  if [ $v = base ]; then export CAPITAL_HIP_LIB=$B/libcapital_hip.so; else unset CAPITAL_HIP_LIB; fi
  timeout -k 10 120 python tools/small_bench.py 128 256 512 2>&1 | grep -v amdgpu | sed "s/^/$v /" | tee -a $O/small.txt
  timeout -k 10 120 python tools/pt_bench.py 2>&1 | grep -v amdgpu | sed "s/^/$v /" | cut -c1-330 | tee -a $O/pt.txt
done
