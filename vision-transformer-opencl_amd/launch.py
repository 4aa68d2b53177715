"""One process per GPU: spawn the rank processes of a single-node job and relay rank 0's output.

The reference initialises exactly one device (ViT_opencl.c:74-101) and loops over its images
(ViT_opencl.c:802); the data-parallel form of that loop is N independent processes, one per GPU, each
forwarding its own shard (dp.py).  `python bench.py --gpus N` must be startable as it stands, so the
PARENT process below creates the ranks itself -- and it has to do so before anything in it has touched
the GPU: a process that has initialised HIP must never fork/exec workers (the children are fresh
interpreters started with subprocess, never a re-exec of this one).  Nothing in this module imports
torch or loads the HIP library.

Environment handed to rank r (the torch.distributed.run contract, so that a script runs the same under
either launcher): RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
from typing import List, Optional, Sequence, Tuple

ENV_MARK = "VIT_LAUNCH_CHILD"   # set in every rank process; a script uses it to tell "launched" from "plain"


def launched() -> bool:
    """True inside a rank process (started by launch_ranks or by torch.distributed.run)."""
    return ENV_MARK in os.environ or "WORLD_SIZE" in os.environ


def rank_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world) of this process; (0, 0, 1) when it was started plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus(timeout: float = 600.0) -> int:
    """GPUs this node shows, counted in a CHILD interpreter: the launching process stays free of torch and of the HIP runtime
    (a process that has touched the GPU must not start workers).  -1 when the count could not be taken."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=timeout)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else -1
    except (subprocess.SubprocessError, ValueError, IndexError, OSError):
        return -1


def _pump(stream, sink, prefix: str, keep: Optional[List[str]]) -> None:
    for line in iter(stream.readline, ""):
        if keep is not None:
            keep.append(line)
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def launch_ranks(script: str, script_args: Sequence[str], nproc: int, timeout: Optional[float] = None,
                 extra_env: Optional[dict] = None, relay_stdout=None) -> Tuple[int, str]:
    """Start `nproc` rank processes `python script *script_args`, wait for all of them.

    Rank 0's stdout is relayed verbatim to `relay_stdout` (default: this process's stdout) and returned;
    every other rank's stdout and every rank's stderr go to this process's stderr with a "[rank r]" prefix.
    Returns (exit code, rank 0 stdout): the exit code is 0 only if every rank exited with 0, otherwise the
    first non-zero one; when a rank fails or the timeout expires, the remaining ranks are terminated
    (each by its own pid -- never by pattern).
    """
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    port = free_port()
    out_sink = relay_stdout if relay_stdout is not None else sys.stdout
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []
    rank0_out: List[str] = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), ENV_MARK: "1",
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        if extra_env:
            env.update({k: str(v) for k, v in extra_env.items()})
        p = subprocess.Popen([sys.executable, "-u", script, *script_args], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        pumps.append(threading.Thread(target=_pump, daemon=True,
                                      args=(p.stdout, out_sink if r == 0 else sys.stderr, "" if r == 0 else f"[rank {r}] ",
                                            rank0_out if r == 0 else None)))
        pumps.append(threading.Thread(target=_pump, args=(p.stderr, sys.stderr, f"[rank {r}] ", None), daemon=True))
    for t in pumps:
        t.start()

    import time
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    pending = set(range(nproc))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
        if pending and (rc != 0 or (deadline is not None and time.monotonic() > deadline)):
            if rc == 0:
                rc = 124  # timeout
            for r in pending:      # one failed rank leaves the others waiting in a barrier: stop exactly those pids
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            pending.clear()
        elif pending:
            time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5)
    return rc, "".join(rank0_out)


def init_process_group(backend: str, device=None):
    """torch.distributed init from the rank environment (nccl = RCCL on GPUs, gloo on CPU); returns (rank, local, world)."""
    import torch.distributed as dist
    rank, local, world = rank_env()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world
