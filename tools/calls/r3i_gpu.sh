#!/bin/bash
# round 3, call i: GPU multirank parity incl. TRSM mode on grids (loopback 2 / 4 ranks), bench.py N = 4 rehearsal with the TRSM extra
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/multirank.log 2>&1; echo "multirank rc=$?" | tee -a $O/summary.txt
tail -4 $O/multirank.log
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPITAL_MULTIPATH_MIN=4096 timeout -k 10 900 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop4.json 2> $O/bench_loop4.err; echo "bench loopback N=4 rc=$?" | tee -a $O/summary.txt
python -m pytest tests/test_gpu_lapack.py -x -q -m gpu -k "geqrf" > $O/lapack.log 2>&1; echo "geqrf tests rc=$?" | tee -a $O/summary.txt
tail -2 $O/lapack.log
cat $O/summary.txt
