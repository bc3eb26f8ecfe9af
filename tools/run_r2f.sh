cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r2f && rm -rf $O && mkdir -p $O
for bc in -5 -4 -3 -2; do timeout -k 10 200 python bench.py --n 32768 --bc $bc --no-qr --no-cpu --steps 4 > $O/n32768_bc$bc.json 2>/dev/null; python -c "
import json,sys; b=json.load(open('$O/n32768_bc$bc.json')); print('n=32768 bc=$bc', round(b['ms_per_step'],2),'ms', round(b['value'],2),'TF/s base', b['config']['base_case_order'], 'res', b['config']['residual'])"; done
for bc in -6 -5 -4 -3; do timeout -k 10 300 python bench.py --n 65536 --bc $bc --no-qr --no-cpu --steps 2 > $O/n65536_bc$bc.json 2>/dev/null; python -c "
import json,sys; b=json.load(open('$O/n65536_bc$bc.json')); print('n=65536 bc=$bc', round(b['ms_per_step'],2),'ms', round(b['value'],2),'TF/s base', b['config']['base_case_order'])"; done
