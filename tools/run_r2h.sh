cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2h && rm -rf $O && mkdir -p $O
timeout -k 10 400 python tools/ts_wide_bench.py 21 23 > $O/blocks.log 2>&1 && CAPI_NO_TALL=1 timeout -k 10 400 python tools/ts_wide_bench.py 21 > $O/off.log 2>&1; tail -6 $O/blocks.log; tail -2 $O/off.log
timeout -k 10 300 python tools/ts_wide_bench.py 20 --n 512 > $O/b512.log 2>&1; tail -2 $O/b512.log
timeout -k 10 300 python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py -m gpu -q -x -k "gram or syrk or cacqr" > $O/tests.log 2>&1; tail -3 $O/tests.log
