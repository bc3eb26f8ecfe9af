# Round-2 evidence, one gpurun call: bench line, the same command under rocprofv3 (kernel stats, marker ranges, timed-region
# average of the roofline kernel), PMC passes (separate runs per counter set) on the headline configuration and on the
# tall-skinny kernels of CholeskyQR2 at n = 256 and n = 1024.   Outputs: gpurun_out/cap2/ (copy what is to be kept into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/cap2 && rm -rf $O && mkdir -p $O
timeout -k 10 600 python bench.py --steps 3 > $O/bench.json 2> $O/bench.err && \
timeout -k 10 900 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 3 --no-cpu > $O/bench_under_rocprof.json 2> $O/rp.err && \
F=$(find $O/tr -name "b_kernel_trace.csv" | head -1) && python tools/timed_region_stats.py $F 3 > $O/timed_region.txt && \
cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv && (cp $(find $O/tr -name "b_marker_api_stats.csv" -o -name "b_marker*stats*.csv" | head -1) $O/marker_stats.csv 2>/dev/null; ls $O/tr/*/ > $O/tr_files.txt 2>&1; rm -rf $O/tr) && \
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do t=$(echo $c | cut -c1-2 | tr A-Z a-z); timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$t -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr --no-config2 > $O/pmc_$t.json 2> $O/pmc_$t.err && grep -E "Counter_Name|dgemm_tile_kernel<128, true, true>" $(find $O/pmc_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_all.csv && python -c "
import json,sys
n=int(json.load(open('$O/pmc_$t.json'))['roofline']['launches_per_step']); c=2 if '$t'=='sq' else 1
l=open('$O/pmc_${t}_all.csv').read().splitlines(); open('$O/pmc_${t}_bench_n65536.csv','w').write('\n'.join(l[:1+n*c])+'\n')   # the factor() launches only (the validator's products follow)
"; rm -rf $O/pmc_$t $O/pmc_${t}_all.csv; done && \
for c in FETCH_SIZE WRITE_SIZE; do t=$(echo $c | cut -c1-2 | tr A-Z a-z); timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ts_$t -o p -- python tools/ts_bench.py 4194304 256 > $O/ts256_$t.log 2>&1 && grep -E "Counter_Name|gram_ts_kernel|trmm_right_ts32_kernel" $(find $O/ts_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_ts256.csv; rm -rf $O/ts_$t; \
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tw_$t -o p -- python tools/ts_bench.py 2097152 1024 > $O/ts1024_$t.log 2>&1 && grep -E "Counter_Name|dgemm_tile_kernel" $(find $O/tw_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_ts1024.csv; rm -rf $O/tw_$t; done
ls -la $O; cat $O/timed_region.txt; cut -c1-400 $O/bench.json
