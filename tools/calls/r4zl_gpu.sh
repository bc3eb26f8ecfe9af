#!/bin/bash
# round 4, call zl: the base-case order of the n = 65536 step once more, now that the diagonal-block routine is faster below 4096: bc_mult -6 (1024, the default), -5 (2048), -4 (4096), -7 (512)
export TMPDIR=/tmp
O=gpurun_out/r4zl; rm -rf $O; mkdir -p $O
for bc in -6 -5 -4 -7 -6 -5; do
  timeout -k 10 200 python bench.py --steps 3 --bc $bc --no-cpu --no-qr --no-config2 2> $O/err_$bc.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bc_mult', $bc, 'base case order', d['config'].get('base_case_order'), 'ms_per_step %.1f' % d['ms_per_step'], 'TFLOP/s %.2f' % d['value'], 'residual %.2e' % d['config']['residual'])
" | tee -a $O/bc.txt
done
