cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2l && rm -rf $O && mkdir -p $O
timeout -k 10 400 python tools/ts_wide_bench.py 21 23 > $O/blocks.log 2>&1 && CAPI_TALL_ONE_LAUNCH=1 timeout -k 10 400 python tools/ts_wide_bench.py 21 23 > $O/one.log 2>&1; tail -5 $O/blocks.log; tail -4 $O/one.log
timeout -k 10 300 python tools/ts_wide_bench.py 20 --n 512 > $O/b512.log 2>&1; tail -2 $O/b512.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/tw -o p -- python tools/ts_bench.py 2097152 1024 > $O/ts1024_fe.log 2>&1 && grep -E "Counter_Name|dgemm_tile_kernel<128, false|trmm_right_ts32" $(find $O/tw -name "p_counter_collection.csv" | head -1) | tail -8 > $O/pmc_fe_trmm.csv; rm -rf $O/tw; awk -F, '{print $9, $(NF-2)}' $O/pmc_fe_trmm.csv | cut -c1-120
timeout -k 10 600 python -m pytest tests/test_gpu_blas.py tests/test_gpu_schedules.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
