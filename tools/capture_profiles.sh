cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/cap && rm -rf $O && mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 300 python bench.py --steps 3 > $O/bench.json 2> $O/bench.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python bench.py --steps 3 > $O/bench_under_rocprof.json 2> $O/rp.err && \
F=$(find $O/tr -name "b_kernel_trace.csv" | head -1) && python tools/timed_region_stats.py $F 3 > $O/timed_region.txt && python tools/step_breakdown.py $F 4 > $O/breakdown.txt && python tools/la_timeline.py $F 4 > $O/la_timeline.txt && cp $(find $O/tr -name "b_kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/tr && \
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do t=$(echo $c | cut -c1-2 | tr A-Z a-z); timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$t -o p -- python bench.py --steps 1 --warmup 0 --no-cpu --no-qr > $O/pmc_$t.json 2> $O/pmc_$t.err && grep -E "Counter_Name|dgemm_tile_kernel<128, true, true>" $(find $O/pmc_$t -name "p_counter_collection.csv" | head -1) > $O/pmc_$t.csv; rm -rf $O/pmc_$t; done
ls -la $O; cat $O/timed_region.txt; cut -c1-300 $O/bench.json
