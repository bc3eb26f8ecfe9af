#!/bin/bash
# round 4, call zd: bench --gpus 4 --one-device over the asynchronous loopback (the config-5 slice divided among the ranks of the one card)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4zd; rm -rf $O; mkdir -p $O
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPI_LOOPBACK_MODE=async timeout -k 10 500 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu > $O/bench_n4_loopback.json 2> $O/bench_n4_loopback.err; echo "bench --gpus 4 --one-device (async loopback) rc=$?" | tee -a $O/legs.txt
cut -c1-3000 $O/bench_n4_loopback.json; grep -v amdgpu $O/bench_n4_loopback.err | tail -5 | cut -c1-400
