#!/bin/bash
# round 3, call q: TRSM mode (one stream, no lookahead) with launches in resident rounds against the default: n = 65536 and 32768
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3q
mkdir -p $O
R="CAPI_ROUNDS=3 CAPI_TRMM_PAIR=2 CAPI_TRMM_PAIR_ROUNDS=1"
for i in 1 2; do
  python tools/pmc_segv_probe.py 65536 3 >> $O/default.log 2>&1
  env $R python tools/pmc_segv_probe.py 65536 3 >> $O/rounds.log 2>&1
  python tools/pmc_segv_probe.py 32768 4 >> $O/default.log 2>&1
  env $R python tools/pmc_segv_probe.py 32768 4 >> $O/rounds.log 2>&1
done
echo default; grep "probe:" $O/default.log; echo rounds; grep "probe:" $O/rounds.log
