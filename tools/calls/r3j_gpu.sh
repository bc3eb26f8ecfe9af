#!/bin/bash
# round 3, call j: the whole GPU suite at the round's kernels and host layer
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3j
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -6 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
tail -2 $O/smoke.log
