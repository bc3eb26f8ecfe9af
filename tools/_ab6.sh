cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab6
for cfg in "0 0" "1 1" "0 0" "1 1"; do set -- $cfg
  CAPI_TILE_ORDER=$1 CAPI_TRMM_PAIR=$2 python bench.py --no-cpu --no-qr --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('order $1 pair $2 n65536', round(d['ms_per_step'],1), round(d['value'],2), 'roof', round(d['roofline']['frac'],4), 'c2', round(d['config2']['ms_per_step'],1), 'trsm', round(d['cholesky_trsm_mode']['ms_per_step'],1))" || exit 1
done > gpurun_out/ab6/time.txt
cat gpurun_out/ab6/time.txt
