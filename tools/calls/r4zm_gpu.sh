#!/bin/bash
# round 4, call zm: the base-case order at n = 32768 (config 2): bc_mult -5 (1024), -4 (2048), -3 (4096)
export TMPDIR=/tmp
O=gpurun_out/r4zm; rm -rf $O; mkdir -p $O
for bc in -5 -4 -3 -5 -4; do
  timeout -k 10 200 python bench.py --n 32768 --steps 5 --bc $bc --no-cpu --no-qr --no-config2 2> $O/err_$bc.txt | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('n=32768 bc_mult', $bc, 'base case order', d['config'].get('base_case_order'), 'ms_per_step %.2f' % d['ms_per_step'], 'TFLOP/s %.2f' % d['value'], 'residual %.2e' % d['config']['residual'])
" | tee -a $O/bc.txt
done
