cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r2m
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2m/tests.log 2>&1; tail -4 gpurun_out/r2m/tests.log
bash tools/capture_profiles_r2.sh
