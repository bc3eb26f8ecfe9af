cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2a && rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
CAPI_TALL_MODE=3 timeout -k 10 300 python tools/ts_wide_bench.py 21 > $O/tall3.log 2>&1; tail -3 $O/tall3.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
