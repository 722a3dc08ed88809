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


def spawn_ranks(world: int, argv: Sequence[str], timeout_s: Optional[float] = None, poll_s: float = 0.05) -> int:
    """Start `world` ranks of `argv` (a full command line, e.g. [sys.executable, "bench.py", "--gpus", "8"]), wait for
    all of them, return 0 if every rank exited 0, else the first non-zero exit code seen.  When one rank fails the
    others are terminated (exact PIDs) instead of being left waiting in a collective."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port)))
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
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
            for r in alive:
                procs[r].send_signal(signal.SIGTERM)
            t_kill = time.monotonic() + 10.0
            for r in sorted(alive):
                try:
                    procs[r].wait(timeout=max(0.0, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            alive.clear()
        if alive:
            time.sleep(poll_s)
    return rc


def relaunch_self(world: int, timeout_s: Optional[float] = None) -> int:
    """spawn_ranks() of the running script with its own arguments."""
    return spawn_ranks(world, [sys.executable] + sys.argv, timeout_s)
