#!/bin/bash
# round 4, call x: the intermittent wrong factor of ch_p3 as the SECOND case of an eight-thread-rank process (asynchronous transport): the sequence
# [ch_p0, ch_p3] with and without delayed receives, with and without multi-path transfers, and with policy 2 in its place
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r4x; rm -rf $O; mkdir -p $O/logs
cat > /tmp/seq.py <<'PY'
import os, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_multirank as T
import oracle as O
O.build()
n = 4096
p0 = {"tag": "ch_p0", "kind": "cholinv", "n": n, "c": 2, "bc": -3, "ci": 1, "serialize": True, "policy": 0}
def p3(policy=3, env=None):
    c = {"tag": "ch_p3", "kind": "cholinv", "n": n + 40, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": policy}
    if env: c["env"] = env
    return c
runs = [("[p0, p3]", [p0, p3()], "0"), ("[p0, p3] delayed", [p0, p3()], "1500"), ("[p0, p3] multipath off", [p0, p3(env={"CAPITAL_MULTIPATH": "0"})], "0"),
        ("[p0, p2]", [p0, p3(2)], "0"), ("[p0, p3] again", [p0, p3()], "0"), ("[p0, p3] delayed again", [p0, p3()], "1500"), ("[p3, p3]", [p3(), dict(p3(), tag="ch_p3b")], "0")]
for name, cases, delay in runs:
    os.environ["CAPI_LOOPBACK_DELAY_US"] = delay
    os.environ["CAPITAL_TEST_RANK_LOG_DIR"] = sys.argv[1]
    with tempfile.TemporaryDirectory() as d:
        try:
            T._launch_thread_ranks(4, 2, {"dir": d, "cases": cases}, "async", timeout=200)
            T._check_cases(O, d, cases, 8, 2)
            print(f"{name}: ok", flush=True)
        except AssertionError as e:
            key = [l for l in str(e).splitlines() if "DriverError" in l or "no device-side" in l or "never published" in l]
            print(f"{name}: FAILED -- {key[:1] if key else str(e)[-300:]}", flush=True)
PY
timeout -k 10 900 python /tmp/seq.py $PWD/$O/logs > $O/seq.txt 2>&1; echo "rc=$?" | tee -a $O/legs.txt
grep -v amdgpu $O/seq.txt | cut -c1-400
