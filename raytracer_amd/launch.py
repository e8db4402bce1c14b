"""One process per GPU without an external launcher.

`bench.py --gpus N` may be started as a plain `python bench.py --gpus N` (no torch.distributed.run around it).  The
parent then starts N fresh child processes of the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set, lets them write to its own stdout / stderr (rank 0 prints the JSON line) and returns the first
non-zero exit status.  The parent never touches the GPU and never replaces itself with another program: the children
are ordinary `subprocess.Popen` processes and the parent waits for them.

Pixels are independent in the reference (Scene.fs:119-121), so ranks only meet in the frame gather (multi.py).
"""
import os
import signal
import socket
import subprocess
import sys
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_as_rank(environ=None):
    """True when this process already is one rank of a job (torch.distributed.run or spawn_ranks set the variables)."""
    return "WORLD_SIZE" in (os.environ if environ is None else environ)


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    return env


def spawn_ranks(argv, world, port=None, grace_s=15.0, poll_s=0.05, timeout_s=None):
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `world` ranks on this node.

    Returns 0 when every rank exits 0, else the exit status of the first rank that failed (a rank killed by signal s
    counts as 128 + s).  When a rank fails, the others get `grace_s` seconds to leave on their own (they are normally
    stuck in a collective with a dead peer) and are then terminated — by PID, never by pattern.
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port() if port is None else port
    # rank 0 owns the parent's stdout (the JSON line); whatever the other ranks print there goes to stderr
    procs = [subprocess.Popen(list(argv), env=rank_env(r, world, port), stdout=None if r == 0 else 2) for r in range(world)]       # (2 = this process's stderr descriptor)
    first_bad, failed_at, t0 = 0, None, time.monotonic()
    try:
        while True:
            running = 0
            for p in procs:
                rc = p.poll()
                if rc is None:
                    running += 1
                elif rc != 0 and first_bad == 0:
                    first_bad = rc if rc > 0 else 128 - rc
                    failed_at = time.monotonic()
            if running == 0:
                break
            if failed_at is not None and time.monotonic() - failed_at > grace_s:
                break
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                first_bad = first_bad or 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return first_bad


def relaunch_self(world, script=None, args=None):
    """`python script args` as `world` ranks; for a script that found --gpus N > 1 and no WORLD_SIZE."""
    script = os.path.abspath(sys.argv[0]) if script is None else script
    args = sys.argv[1:] if args is None else args
    return spawn_ranks([sys.executable, "-u", script] + list(args), world)
