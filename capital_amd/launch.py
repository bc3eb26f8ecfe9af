"""Process launcher: what `mpiexec -n P ./cholinv ...` does for the reference's benches (bench/cholesky/cholinv.cpp:8-13,
SURVEY.md App. A) -- one fresh process per GPU of ONE node, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
environment, rank 0's stdout relayed, the first non-zero exit propagated and the other ranks ended, a bounded wall time.

The launching process never touches a GPU (it counts devices with `torch.cuda.device_count()`, which does not initialise
the runtime on this image), so it neither holds a slot on a card nor has to re-exec: the ranks are ordinary children.
"""
import os
import signal
import socket
import subprocess
import sys
import time

PARENT_ENV = "CAPITAL_LAUNCH_PARENT"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    import torch
    return torch.cuda.device_count()


def die_with_parent():
    """Called first thing by a launched rank: if the launcher goes away (its own limit, a signal from the driver), the rank
    must not stay behind on the GPU.  PR_SET_PDEATHSIG = 1."""
    ppid = os.environ.get(PARENT_ENV)
    if not ppid:
        return
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)
    except OSError:
        pass
    if os.getppid() != int(ppid):      # the launcher died between fork and prctl
        os._exit(1)


def _end(procs, grace_s=10.0):
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + grace_s
    for p in procs:
        while p.poll() is None and time.monotonic() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
            p.wait()


def run_ranks(nranks, argv, timeout_s=1800.0, one_device=False, extra_env=None, out=None, err=None):
    """Start `argv` nranks times, rank r on GPU r (all on GPU 0 with one_device: the loopback rehearsal of tests/rccl_loopback).
    Returns the exit code of the job: 0 if every rank returned 0, else the first non-zero code seen (124 on timeout)."""
    out = out if out is not None else sys.stdout
    err = err if err is not None else sys.stderr
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_device else str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env[PARENT_ENV] = str(os.getpid())
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")
        env.setdefault("CAPITAL_RUN_ID", f"{os.getpid()}_{port}")
        if extra_env:
            env.update(extra_env)
        # rank 0's stdout IS the job's stdout (the one JSON line); the other ranks' prints go to stderr
        procs.append(subprocess.Popen(argv, env=env, stdout=out if r == 0 else err, stderr=err))
    t_end = time.monotonic() + timeout_s
    code = 0
    try:
        live = set(range(nranks))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    print(f"launch: rank {r} exited with {rc}; ending the other {len(live)} rank(s)", file=err, flush=True)
            if code != 0:
                break
            if time.monotonic() > t_end:
                print(f"launch: {len(live)} rank(s) still running after {timeout_s:.0f} s; ending them", file=err, flush=True)
                code = 124
                break
            time.sleep(0.05)
    finally:
        _end(procs)
    return code


def self_launch(nranks, script, args, timeout_s=1800.0, one_device=False, check_devices=True):
    """`python bench.py --gpus N` without a launcher around it: refuse at once if the node has fewer devices, else run N ranks."""
    if check_devices and not one_device:
        have = visible_gpus()
        if have < nranks:
            print(f"{os.path.basename(script)}: --gpus {nranks} needs {nranks} GPUs, this node shows {have}", file=sys.stderr, flush=True)
            return 2
    return run_ranks(nranks, [sys.executable, script] + list(args), timeout_s=timeout_s, one_device=one_device)
