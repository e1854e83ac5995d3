"""One process per GPU on one node, for `python bench.py --gpus N` run by itself.

The driver starts multi-GPU runs with torch.distributed.run; this is the same thing for a user who types the
command README.md documents.  The ranks are fresh child processes started BEFORE anything in this process has
touched the GPU (a process that has initialised HIP must never be replaced or forked into workers), each with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set; rank 0's stdout is passed through (the one JSON
line), every rank's stderr is passed through, and the exit code is the first non-zero one.
"""
from __future__ import annotations

import os
import socket
import subprocess
import time
from typing import Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return env


def launch_ranks(world: int, argv: Sequence[str], timeout: float = None) -> int:
    """Run `argv` as `world` ranks; returns the exit code (0 only if every rank returned 0)."""
    port = free_port()
    procs = []
    for rank in range(world):
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=rank_env(rank, world, port), stdout=out))
    code = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        live = list(procs)
        while live and code == 0:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    code = rc  # one rank failed: the others would wait in a collective for ever
                    break
            else:
                if deadline is not None and time.monotonic() > deadline:
                    code = 124
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=10)
            except subprocess.TimeoutExpired:
                q.kill()
    return code
