cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/tsw1 && rm -rf $O && mkdir -p $O
timeout -k 10 500 python tools/ts_wide_bench.py 21 23 > $O/tall.log 2>&1 && \
CAPI_NO_TALL=1 timeout -k 10 400 python tools/ts_wide_bench.py 21 > $O/notall.log 2>&1 && \
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python tools/ts_wide_bench.py 21 > $O/pmc_$c.log 2>&1 && grep -E "Counter_Name|dgemm_tile_kernel" $(find $O/pmc_$c -name "p_counter_collection.csv" | head -1) | tail -12 > $O/pmc_$c.csv; rm -rf $O/pmc_$c; done
tail -8 $O/tall.log; tail -3 $O/notall.log
