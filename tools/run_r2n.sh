cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2n && rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lapack.py tests/test_gpu_engine_abi.py -m gpu -q > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout -k 10 300 python tools/geqrf_bench.py > $O/geqrf.log 2>&1; tail -3 $O/geqrf.log
