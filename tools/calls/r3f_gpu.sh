#!/bin/bash
# round 3, call f: n = 65536 full-size test, bench.py N = 2 / 4 rehearsal (loopback transport) with the communication-form probes, the headline line
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k n65536 > $O/fullsize.log 2>&1; echo "fullsize n65536 rc=$?" | tee -a $O/summary.txt
tail -3 $O/fullsize.log
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPITAL_MULTIPATH_MIN=4096 timeout -k 10 900 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop4.json 2> $O/bench_loop4.err; echo "bench loopback N=4 rc=$?" | tee -a $O/summary.txt
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so timeout -k 10 900 python bench.py --gpus 2 --one-device --n 8192 --steps 2 --no-cpu --no-qr > $O/bench_loop2.json 2> $O/bench_loop2.err; echo "bench loopback N=2 rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt
