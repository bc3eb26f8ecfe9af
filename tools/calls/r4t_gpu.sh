#!/bin/bash
# round 4, call t: the row panel by a strip solve (leaf only factors, all block inverses in one batched launch): parity, then A/B against the TRMM form
export TMPDIR=/tmp
O=gpurun_out/r4t; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lapack.py tests/test_gpu_schedules.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -3 $O/pytest.log
for v in 1 0 1 0; do echo "CAPI_PANEL_SOLVE=$v" >> $O/pt.txt; CAPI_PANEL_SOLVE=$v python tools/pt_bench.py 2>&1 | grep -v amdgpu >> $O/pt.txt; done
cat $O/pt.txt
for v in 1 0 1 0; do
  CAPI_PANEL_SOLVE=$v timeout -k 10 200 python bench.py --n 32768 --steps 6 --no-cpu --no-qr > $O/b.json 2> $O/b.err
  python - >> $O/ab.txt <<PY
import json
j = json.loads([l for l in open("$O/b.json") if l.startswith("{")][-1])
print("CAPI_PANEL_SOLVE=$v: n=32768 %.2f ms/step, residual %.2e" % (j["ms_per_step"], j["config"]["residual"]))
PY
done
cat $O/ab.txt
