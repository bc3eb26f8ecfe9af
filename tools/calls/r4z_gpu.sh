#!/bin/bash
# round 4, call z: (1) do IPC mappings of SMALL hipMalloc blocks show the exporter's bytes?  (2) the failing sequence [ch_p0, ch_p3] on eight thread-ranks (async)
# with rings cut at 1 MiB (as before) and at 4 MiB (never pieces of a shared block; they still grow)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export O=gpurun_out/r4z; rm -rf $O; mkdir -p $O/logs
for spec in "1048576 32" "65536 64" "4194304 16" "1048576 128"; do
  echo "=== ipc_fragment_probe $spec" >> $O/probe.txt
  timeout -k 10 120 tests/rccl_loopback/ipc_fragment_probe $spec >> $O/probe.txt 2>&1; echo "probe $spec rc=$?" | tee -a $O/legs.txt
done
cut -c1-400 $O/probe.txt | head -60
sed -e 's/for ring in 1 1 1; do/for ring in 4 4 1; do/' -e 's#gpurun_out/r4y#gpurun_out/r4z#' -e 's#^O=.*#O=gpurun_out/r4z#' tools/calls/r4y_gpu.sh | sed -n '/^cat > \/tmp\/where.py/,$p' > /tmp/rest.sh
bash /tmp/rest.sh
