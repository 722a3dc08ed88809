"""One process per GPU without an external launcher.

`python bench.py --gpus N` with no WORLD_SIZE in the environment calls spawn_ranks(): it starts N copies of the same
command, one per rank, with the variables torch.distributed.run would set (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT), waits for them and returns the worst exit code.  The parent never touches a GPU (no torch import, no HIP
call): the ranks are plain child processes, started before anything initialises a device.

When the driver launches the ranks itself (python -m torch.distributed.run ...) WORLD_SIZE is already set and this
module is not used.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _pdeathsig() -> None:
    """Child side (preexec): SIGTERM when the parent goes away, however it went (prctl PR_SET_PDEATHSIG = 1)."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)
    except Exception:   # not Linux: the try/finally of spawn_ranks still covers every exit the parent takes itself
        pass


def _stop(procs: Sequence[subprocess.Popen], grace_s: float = 10.0) -> None:
    """SIGTERM, then SIGKILL, to exactly these children (never by pattern)."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.send_signal(signal.SIGTERM)
        except ProcessLookupError:
            pass
    t_kill = time.monotonic() + grace_s
    for p in live:
        try:
            p.wait(timeout=max(0.0, t_kill - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


class _Terminated(Exception):
    pass


def spawn_ranks(world: int, argv: Sequence[str], timeout_s: Optional[float] = None, poll_s: float = 0.05) -> int:
    """Start `world` ranks of `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "8"]), wait for
    all of them, return 0 if every rank exited 0, else the first non-zero exit code seen.  When one rank fails the
    others are terminated (exact PIDs) instead of being left waiting in a collective.  The ranks never outlive this
    process: a SIGTERM / SIGINT / KeyboardInterrupt here stops them on the way out (try/finally), and a parent that is
    killed outright takes them along through the parent-death signal."""
    if world < 1:
        raise ValueError("world must be >= 1")
    procs: List[subprocess.Popen] = []
    rc = 0

    def on_term(signum, _frame):
        raise _Terminated(signum)

    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
    except ValueError:   # not the main thread: signals cannot be redirected from here, finally still runs on exceptions
        old_term = None
    try:
        # the port is released before rank 0 binds it, so another job on this host may take it in between: a run whose
        # rendezvous fails for that reason (every rank exits non-zero within seconds) is started again on a new port
        for attempt in range(3):
            port = free_port()
            procs = [subprocess.Popen(list(argv), env=rank_env(r, world, port), preexec_fn=_pdeathsig) for r in range(world)]
            t_start = time.monotonic()
            deadline = None if timeout_s is None else t_start + timeout_s
            rc = 0
            alive = set(range(world))
            while alive:
                for r in sorted(alive):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    alive.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                failed = rc != 0 or (deadline is not None and time.monotonic() > deadline)
                if failed and alive:
                    if rc == 0:
                        rc = 124  # timeout
                    _stop([procs[r] for r in alive])
                    alive.clear()
                if alive:
                    time.sleep(poll_s)
            port_taken = rc not in (0, 124) and world > 1 and time.monotonic() - t_start < 20.0 and _port_in_use(port)
            if not port_taken:
                break
        return rc
    except _Terminated as t:
        return 128 + int(t.args[0])
    finally:
        _stop(procs)
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)


def _port_in_use(port: int) -> bool:
    """True if somebody else is listening on 127.0.0.1:port now (our own ranks have exited when this is asked)."""
    with socket.socket() as s:
        try:
            s.bind(("127.0.0.1", port))
            return False
        except OSError:
            return True


def relaunch_self(world: int, timeout_s: Optional[float] = None) -> int:
    """spawn_ranks() of the running script with its own arguments."""
    return spawn_ranks(world, [sys.executable] + sys.argv, timeout_s)
