cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2j && rm -rf $O && mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
b=json.load(open('gpurun_out/r2j/bench.json'))
print('value',round(b['value'],2),'ms',round(b['ms_per_step'],1),'roofline',round(b['roofline']['frac'],4))
for k in ('config2','cholesky_trsm_mode','cacqr2','cacqr2_config5'): print(k, round(b[k]['tflops'],2), round(b[k].get('ms',b[k].get('ms_per_step')),2))
PY
for c in FETCH_SIZE WRITE_SIZE; do t=$(echo $c | cut -c1-2 | tr A-Z a-z); timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/tw_$t -o p -- python tools/ts_bench.py 2097152 1024 > $O/ts1024_$t.log 2>&1 && grep -E "Counter_Name|dgemm_tile_kernel|gram_ts_kernel" $(find $O/tw_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_${t}_ts1024.csv; rm -rf $O/tw_$t; done; tail -1 $O/ts1024_fe.log
