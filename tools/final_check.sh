# HEAD check on a GPU box: the driver's own sequence (GPU tests with -x, smoke, default bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/final && rm -rf $O && mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -2 $O/bench.err
