#!/bin/bash
# round 4, call r: the intermittent wrong factor of case ch_p3 (NoReplicationOverlap on the 2 x 2 x 2 cube, eight thread-ranks, asynchronous transport):
# made deterministic with delayed receives, and bisected over policy / padding / multi-path
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4r; rm -rf $O; mkdir -p $O
cat > /tmp/bisect.py <<'PY'
import os, sys, tempfile, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_multirank as T
import oracle as O
O.build()
n = 4096
variants = {
  "p3 padded (the failing case)": {"tag": "v", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 3},
  "p3 padded, multipath off": {"tag": "v", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 3, "env": {"CAPITAL_MULTIPATH": "0"}},
  "p2 padded (no overlap)": {"tag": "v", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 2},
  "p3 unpadded": {"tag": "v", "kind": "cholinv", "n": n, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 3},
  "p0 padded": {"tag": "v", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 0},
}
for name, case in variants.items():
    for delay in ("0", "1500"):
        os.environ["CAPI_LOOPBACK_DELAY_US"] = delay
        os.environ["CAPITAL_TEST_RANK_LOG_DIR"] = sys.argv[1]
        with tempfile.TemporaryDirectory() as d:
            try:
                T._launch_thread_ranks(4, 2, {"dir": d, "cases": [case]}, "async", timeout=200)
                T._check_cases(O, d, [case], 8, 2)
                print(f"{name}, receive delay {delay} us: ok", flush=True)
            except AssertionError as e:
                msg = str(e)
                key = [l for l in msg.splitlines() if "DriverError" in l or "no device-side" in l or "never published" in l]
                print(f"{name}, receive delay {delay} us: FAILED -- {key[:2] if key else msg[-300:]}", flush=True)
PY
mkdir -p $O/logs
timeout -k 10 900 python /tmp/bisect.py $PWD/$O/logs > $O/bisect.txt 2>&1; echo "bisect rc=$?" | tee -a $O/legs.txt
grep -v amdgpu $O/bisect.txt | cut -c1-500
