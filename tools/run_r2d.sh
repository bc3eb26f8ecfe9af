cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2d && rm -rf $O && mkdir -p $O
timeout -k 10 200 python tools/flush_probe.py > $O/flush.log 2>&1; cat $O/flush.log
timeout -k 10 400 python tools/ts_wide_bench.py 21 > $O/tall_dense.log 2>&1 && CAPI_NO_TALL=1 timeout -k 10 400 python tools/ts_wide_bench.py 21 > $O/tall_off.log 2>&1; tail -3 $O/tall_dense.log; tail -2 $O/tall_off.log
timeout -k 10 300 python tools/ts_wide_bench.py 20 --n 512 > $O/tall512_dense.log 2>&1 && CAPI_NO_TALL=1 timeout -k 10 300 python tools/ts_wide_bench.py 20 --n 512 > $O/tall512_off.log 2>&1; tail -2 $O/tall512_dense.log; tail -2 $O/tall512_off.log
