#!/bin/bash
# round 4, call zg: the leaf's pivot check without a branch: parity (incl. the info word), leaf trace, order 512..8192
export TMPDIR=/tmp
O=gpurun_out/r4zg; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_lapack.py tests/test_golden.py tests/test_gpu_schedules.py -m gpu -x -q > $O/lapack.log 2>&1; echo "lapack rc=$?" | tee -a $O/legs.txt; tail -3 $O/lapack.log
grep -q "lapack rc=0" $O/legs.txt || exit 1
for v in 1 2; do timeout -k 10 120 python tools/pt_bench.py 2>&1 | grep -v amdgpu | tee -a $O/pt_bench.txt; done
CAPI_LEAF_TRACE=1 timeout -k 10 120 python tools/leaf_bench.py > $O/leaf_trace.txt 2>&1; grep "leaf b=128" $O/leaf_trace.txt | head -3; tail -3 $O/leaf_trace.txt | cut -c1-300
