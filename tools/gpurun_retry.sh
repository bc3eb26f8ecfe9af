#!/bin/bash
# gpurun with patience: status "transient" means "no box or slot free, nothing charged" -- wait and ask again (a command that RAN is never retried)
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if ! python3 -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/.last_call.json')).get('status') == 'transient' else 1)" 2>/dev/null; then exit $rc; fi
  echo "[gpurun_retry] no slot (attempt $i), waiting 120 s"
  sleep 120
done
exit 3
