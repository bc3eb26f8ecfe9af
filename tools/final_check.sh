#!/bin/bash
# what the driver runs at the end of a round, in one gpurun call: the GPU suite, smoke(), the default bench line
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/legs.txt; tail -32 $O/pytest.log | cut -c1-200
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/legs.txt; tail -3 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/legs.txt
cut -c1-1500 $O/bench.json
