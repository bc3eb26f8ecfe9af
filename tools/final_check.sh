cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/final && rm -rf $O && mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
