cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/tsw2 && rm -rf $O && mkdir -p $O
TSW_EXTRA=1 timeout -k 10 500 python tools/ts_wide_bench.py 21 > $O/mode2.log 2>&1 && \
CAPI_TALL_MODE=0 timeout -k 10 400 python tools/ts_wide_bench.py 21 > $O/mode0.log 2>&1 && \
for md in 2 0; do CAPI_TALL_MODE=$md timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$md -o p -- python tools/ts_wide_bench.py 21 > $O/pmc_$md.log 2>&1 && grep -E "Counter_Name|dgemm_tile_kernel" $(find $O/pmc_$md -name "p_counter_collection.csv" | head -1) | tail -5 > $O/pmc_fetch_mode$md.csv; rm -rf $O/pmc_$md; done
tail -6 $O/mode2.log; tail -3 $O/mode0.log; cut -c1-60,250-400 $O/pmc_fetch_mode*.csv
