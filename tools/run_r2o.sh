cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2o && rm -rf $O && mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lapack.py tests/test_gpu_engine_abi.py -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout -k 10 300 python tools/geqrf_bench.py > $O/geqrf.log 2>&1; tail -3 $O/geqrf.log
CAPI_GEQRF_NO_RECONSTRUCT=1 timeout -k 10 300 python tools/geqrf_bench.py > $O/geqrf_old.log 2>&1; tail -3 $O/geqrf_old.log
