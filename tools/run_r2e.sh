cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_gpu_schedules.py tests/test_gpu_blas.py -m gpu -q -x > gpurun_out/r2e/tests.log 2>&1; tail -4 gpurun_out/r2e/tests.log
bash tools/capture_profiles_r2.sh
