#!/bin/bash
# what the driver runs at the end of a round, in one gpurun call: the GPU suite, smoke(), the default bench line
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -32 $O/pytest.log | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/legs.txt; tail -3 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/legs.txt
cut -c1-1500 $O/bench.json
# the N > 1 bench path on this one GPU, four ranks over the asynchronous loopback transport (the rehearsal INTEGRATION.md names)
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPI_LOOPBACK_MODE=async HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu > $O/bench_n4_loopback.json 2> $O/bench_n4_loopback.err; echo "bench --gpus 4 --one-device (async loopback) rc=$?" | tee -a $O/legs.txt
cut -c1-700 $O/bench_n4_loopback.json
