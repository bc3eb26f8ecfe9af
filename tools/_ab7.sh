cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab7
run() { python bench.py --no-cpu --no-qr --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1 n65536', round(d['ms_per_step'],1), round(d['value'],2), 'roof', round(d['roofline']['frac'],4), 'maxlaunch', round(d['roofline']['max_launch_ms'],1), 'c2', round(d['config2']['ms_per_step'],1), 'trsm', round(d['cholesky_trsm_mode']['ms_per_step'],1), 'peak', round(d['mfma_f64_loop_tflops'],2))"; }
for i in 1 2; do
  CAPITAL_HIP_LIB=$PWD/_ab_old/libcapital_hip.so LD_LIBRARY_PATH=$PWD/_ab_old:$LD_LIBRARY_PATH run old || exit 1
  run new || exit 1
done > gpurun_out/ab7/time.txt
cat gpurun_out/ab7/time.txt
