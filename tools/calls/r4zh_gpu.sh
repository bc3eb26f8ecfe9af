#!/bin/bash
# round 4, call zh: the leaf's factor loop without workgroup barriers (CAPI_LEAF_FLOW=1) against the barrier form (=0): parity, order 512..8192, phase trace
export TMPDIR=/tmp
O=gpurun_out/r4zh; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_lapack.py tests/test_golden.py -m gpu -x -q > $O/lapack.log 2>&1; echo "lapack rc=$?" | tee -a $O/legs.txt; tail -3 $O/lapack.log | cut -c1-300
grep -q "lapack rc=0" $O/legs.txt || exit 1
for v in 0 1 0 1; do CAPI_LEAF_FLOW=$v timeout -k 10 120 python tools/pt_bench.py 2>&1 | grep -v amdgpu | sed "s/^/CAPI_LEAF_FLOW=$v /" | tee -a $O/pt_bench.txt; done
for v in 0 1; do CAPI_LEAF_FLOW=$v CAPI_LEAF_TRACE=1 timeout -k 10 120 python tools/leaf_bench.py > $O/leaf_trace_$v.txt 2>&1; grep "leaf b=128" $O/leaf_trace_$v.txt | head -2; tail -1 $O/leaf_trace_$v.txt | cut -c1-300; done
