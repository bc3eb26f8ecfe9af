cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2b && rm -rf $O && mkdir -p $O
cat /sys/fs/cgroup/cpu.max > $O/cpu.txt 2>&1; nproc >> $O/cpu.txt; python -c "import torch; print(torch.cuda.device_count())" >> $O/cpu.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -5 $O/bench.err
