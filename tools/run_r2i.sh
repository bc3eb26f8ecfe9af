cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2i && rm -rf $O && mkdir -p $O
CAPI_DEBUG_GEMM=1 timeout -k 10 400 python tools/ts_wide_bench.py 23 > $O/blocks.log 2> $O/dbg.log; tail -3 $O/blocks.log; grep "M=256 N=256 K=8388608" $O/dbg.log | sort | uniq -c | head -5
timeout -k 10 300 python tools/ts_wide_bench.py 20 --n 512 > $O/b512.log 2>&1; tail -2 $O/b512.log
timeout -k 10 600 python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
