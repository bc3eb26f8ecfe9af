#!/bin/bash
# round 3, first GPU call: N > 1 on one GPU through the loopback transport, bench.py's own launcher, the --pmc SIGSEGV probe
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/multirank.log 2>&1; echo "multirank rc=$?" | tee -a $O/summary.txt
tail -5 $O/multirank.log
python bench.py --gpus 2 > $O/refuse.out 2> $O/refuse.err; echo "bench --gpus 2 on 1 GPU rc=$? (expect 2)" | tee -a $O/summary.txt
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so timeout -k 10 600 python bench.py --gpus 2 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop2.json 2> $O/bench_loop2.err; echo "bench loopback N=2 rc=$?" | tee -a $O/summary.txt
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so timeout -k 10 600 python bench.py --gpus 4 --one-device --n 8192 --steps 2 --no-cpu --qr-rows 65536 > $O/bench_loop4.json 2> $O/bench_loop4.err; echo "bench loopback N=4 rc=$?" | tee -a $O/summary.txt
CAPI_RCCL_LIB=$PWD/tests/rccl_loopback/librccl_loopback.so CAPITAL_BENCH_CHUNKS=4 timeout -k 10 600 python bench.py --gpus 2 --one-device --n 8192 --steps 2 --no-cpu --no-qr > $O/bench_loop2c.json 2> $O/bench_loop2c.err; echo "bench loopback N=2 chunks rc=$?" | tee -a $O/summary.txt
# the SIGSEGV probe: today's default path first, then the call path of the round-2 record
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_new -- python tools/pmc_segv_probe.py 32768 2 > $O/probe_new.out 2> $O/probe_new.err; echo "pmc probe (blocked diag) rc=$?" | tee -a $O/summary.txt
CAPI_POTRF_DIAG=rec CAPI_DEBUG_GEMM=1 timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rec -- python tools/pmc_segv_probe.py 32768 2 > $O/probe_rec.out 2> /tmp/probe_rec.err; echo "pmc probe (recursive diag, round-2 path) rc=$?" | tee -a $O/summary.txt
grep -v "capi gemm" /tmp/probe_rec.err | tail -80 > $O/probe_rec.err
grep "capi gemm" /tmp/probe_rec.err | tail -40 > $O/probe_rec_last_launches.txt
grep -c "small launch" /tmp/probe_rec.err > $O/probe_rec_small_launch_count.txt
rm -rf $O/pmc_new/*/*.db $O/pmc_rec/*/*.db 2>/dev/null
du -sh $O
cat $O/summary.txt
