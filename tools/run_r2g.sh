cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2g && rm -rf $O && mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout -k 10 300 python bench.py --no-qr --no-cpu --no-config2 --steps 2 > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json; tail -3 $O/bench.err
