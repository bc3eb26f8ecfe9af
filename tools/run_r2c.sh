cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2c && rm -rf $O && mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -15 $O/tests.log
