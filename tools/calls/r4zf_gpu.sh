#!/bin/bash
# round 4, call zf: the row panel formed inside the leaf launch (one device-side hand-off instead of a dependent launch): parity, then order 512..8192 timings A/B
export TMPDIR=/tmp
O=gpurun_out/r4zf; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_lapack.py -m gpu -x -q > $O/lapack.log 2>&1; echo "lapack rc=$?" | tee -a $O/legs.txt; tail -3 $O/lapack.log
grep -q "lapack rc=0" $O/legs.txt || exit 1
for v in 0 1 0 1; do CAPI_LEAF_PANEL=$v timeout -k 10 120 python tools/pt_bench.py 2>&1 | grep -v amdgpu | sed "s/^/CAPI_LEAF_PANEL=$v /" | tee -a $O/pt_bench.txt; done
timeout -k 10 200 python -m pytest tests/test_gpu_schedules.py tests/test_golden.py -m gpu -x -q > $O/sched.log 2>&1; echo "schedules rc=$?" | tee -a $O/legs.txt; tail -3 $O/sched.log
