#!/bin/bash
# round 4, call y: WHERE is the second case wrong?  [ch_p0, ch_p3] on eight thread-ranks (async), per rank: error map of R against the oracle
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4y; rm -rf $O; mkdir -p $O/logs
cat > /tmp/where.py <<'PY'
import os, sys, tempfile
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_multirank as T
import oracle as O
O.build()
n = 4096
p0 = {"tag": "ch_p0", "kind": "cholinv", "n": n, "c": 2, "bc": -3, "ci": 1, "serialize": True, "policy": 0}
p3 = {"tag": "ch_p3", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 3}
os.environ["CAPITAL_TEST_RANK_LOG_DIR"] = sys.argv[1]
for attempt in range(4):
    with tempfile.TemporaryDirectory() as d:
        try:
            T._launch_thread_ranks(4, 2, {"dir": d, "cases": [p0, p3]}, "async", timeout=200)
        except AssertionError as e:
            print("attempt", attempt, "ranks failed"); break
        nn = p3["n"]
        A = O.distribute_symmetric(nn, nn, 0, 0, 1, 1)
        Rref, Iref, info = O.cholinv_factor(A, p3["ci"], 1, p3["bc"], 2, 2)
        bad = False
        for r in range(8):
            z = np.load(os.path.join(d, f"ch_p3_rank{r}.npz"))
            x, y, zz, dd, cc = [int(v) for v in z["xyz"]]
            ref = O.cyclic_extract(Rref, x, y, dd, dd)
            err = np.abs(z["R"] - ref)
            tol = 1e-12 * np.abs(Rref).max()
            if err.max() > tol:
                bad = True
                rows, cols = np.where(err > tol)
                print(f"attempt {attempt}: rank {r} (x={x} y={y} z={zz}) residual {float(z['residual']):.2e}: {len(rows)} wrong entries of {err.size}; rows {rows.min()}..{rows.max()}, cols {cols.min()}..{cols.max()}; "
                      f"first wrong col {cols.min()} has wrong rows {sorted(set(rows[cols == cols.min()]))[:6]}...; max err {err.max():.2e}; distinct wrong cols {len(set(cols))}")
            else:
                print(f"attempt {attempt}: rank {r} (x={x} y={y} z={zz}) ok, residual {float(z['residual']):.2e}")
        if bad:
            break
PY
for ring in 1 1 1; do
  echo "=== CAPI_LOOPBACK_MIN_RING_MB=$ring" >> $O/where.txt
  CAPI_LOOPBACK_MIN_RING_MB=$ring timeout -k 10 400 python /tmp/where.py $PWD/$O/logs >> $O/where.txt 2>&1; echo "ring $ring rc=$?" | tee -a $O/legs.txt
done
grep -v amdgpu $O/where.txt | cut -c1-500 | grep -v "ok, residual"
grep -h "DATA CHECK\|no device-side" $O/logs/*.log | cut -c1-600 | head

