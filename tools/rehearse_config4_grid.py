"""Configs 4 and 5 in their own rank counts at a size that means something, on ONE MI355X: the 2 x 2 x 2 grid as eight thread-ranks (four processes x two threads, tests/thread_ranks)
over the asynchronous loopback transport, n = 16384 by default (every rank's block 8192 x 8192; n = 65536 itself needs eight cards' HBM).  No oracle at this
size: the reference validator's residual on the grid, and the two depth layers -- which hold the same blocks -- must agree bit for bit in the sums of R.
    python tools/rehearse_config4_grid.py [n] [out.txt]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_multirank as T      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cases = [
        {"tag": "plain", "kind": "cholinv", "n": n, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 0, "light": True},
        {"tag": "chunks4", "kind": "cholinv", "n": n, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 0, "chunks": 4, "light": True},
        {"tag": "multipath_chunks4", "kind": "cholinv", "n": n, "c": 2, "bc": -2, "ci": 0, "serialize": True, "policy": 3, "chunks": 4, "light": True,
         "env": {"CAPITAL_MULTIPATH": "1"}},
        {"tag": "trsm_mode", "kind": "cholinv", "n": n, "c": 2, "bc": -2, "ci": 0, "serialize": False, "policy": 0, "trsm": True, "light": True},
    ]
    only = os.environ.get("REHEARSE_ONLY")
    if only:
        cases = [c for c in cases if c["tag"] in only.split(",")]
    # one launch per case: the rehearsal's transport keeps its rings to the end of a process (hipFree would drain a sibling rank's stream), and at this size a case's
    # rings are tens of GiB
    ok = True
    print(f"2 x 2 x 2 grid, n = {n}, eight thread-ranks (four processes x two threads) on one GPU over the asynchronous loopback transport", file=out)
    for case in cases:
        with tempfile.TemporaryDirectory() as d:
            t0 = time.time()
            T._launch_thread_ranks(4, 2, {"dir": d, "cases": [case]}, "async", timeout=400)
            wall = time.time() - t0
            z = [np.load(os.path.join(d, f"{case['tag']}_rank{r}.npz")) for r in range(8)]
            res = max(float(v["residual"]) for v in z)
            by_xy = {}
            for v in z:
                x, y, zz = [int(q) for q in v["xyz"][:3]]
                by_xy.setdefault((x, y), []).append(v["sums"])
            layers_agree = all(len(s) == 2 and np.array_equal(s[0], s[1]) for s in by_xy.values())
            secs = max(float(v["seconds"][0]) for v in z), max(float(v["seconds"][1]) for v in z)
            good = res <= 1e-14 and layers_agree
            ok &= good
            print(f"  {case['tag']:>18}: validator residual {res:.2e}; the depth layers' sums of R are {'bit-identical' if layers_agree else 'DIFFERENT'}; "
                  f"factor() {secs[0]:.2f} s then {secs[1]:.2f} s incl. the residual, processes {wall:.0f} s (eight ranks share the card: not a rate)  {'ok' if good else 'FAILED'}", file=out)
            out.flush()
    # config 5's shape, CA-CholeskyQR2 with width 1024 in 1-D row blocks over the eight ranks (2^18 rows each here, 2^23 on eight cards): R is replicated -- every
    # rank must hold the same bits -- and the validators' residual / orthogonality are the reference's
    m_loc = 1 << 18
    case = {"tag": "cacqr2_w1024", "kind": "cacqr", "m": m_loc * 8, "n": 1024, "serialize": False, "light": True}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        T._launch_thread_ranks(4, 2, {"dir": d, "cases": [case]}, "async", timeout=400)
        wall = time.time() - t0
        z = [np.load(os.path.join(d, f"{case['tag']}_rank{r}.npz")) for r in range(8)]
        res, orth = max(float(v["residual"]) for v in z), max(float(v["orth"]) for v in z)
        same = all(np.array_equal(v["sums"], z[0]["sums"]) for v in z)
        good = res <= 1e-13 and orth <= 1e-13 and same
        ok &= good
        print(f"CA-CholeskyQR2 m = {m_loc * 8} (8 x {m_loc} rows), n = 1024, the same eight ranks: residual {res:.2e}, orthogonality {orth:.2e}; R is "
              f"{'bit-identical on all eight ranks' if same else 'NOT the same on all ranks'}; processes {wall:.0f} s  {'ok' if good else 'FAILED'}", file=out)
    print("all ok" if ok else "FAILED", file=out)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
