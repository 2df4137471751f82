"""Start the ranks of a tensor-/pipeline-parallel job from one command.

The reference's `LLM(tp_size=N)` spawns one worker process per GPU itself (nanovllmx/core/executor.py:32-49,
core/worker_client.py:44-49) and ships a pickled batch to each of them every step.  Here every rank runs the same
deterministic engine (SPMD), so all a launcher has to do is start N copies of the caller's script with the
rendezvous variables set: `run_ranks()` does that through `torch.distributed.run`, as a FRESH CHILD PROCESS of a
parent that has not touched the GPU (a process that has initialised HIP must never replace itself with another
program, and need not: the parent only waits and hands the children's exit code on).

Nothing in this module imports torch or loads the HIP library.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Optional, Sequence

RENDEZVOUS_VARS = ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def launcher_env_present(env=None) -> bool:
    """True inside a rank that a launcher started (torch.distributed.run or run_ranks below)."""
    env = os.environ if env is None else env
    return all(v in env for v in RENDEZVOUS_VARS)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_launch_command(nproc: int, script: str, script_args: Sequence[str] = (), port: Optional[int] = None,
                        python: Optional[str] = None) -> list:
    """The command that starts `nproc` ranks of `script` on this node, one per GPU, rendezvous on 127.0.0.1
    (the container's host name may not resolve)."""
    if nproc < 1:
        raise ValueError(f"cannot start {nproc} ranks")
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *script_args]


def run_ranks(nproc: int, script: str, script_args: Sequence[str] = (), env=None, port: Optional[int] = None) -> int:
    """Run `script` as `nproc` ranks and return the launcher's exit code.  The children inherit stdout/stderr, so
    whatever rank 0 prints (bench.py's JSON line) appears as if this process had printed it."""
    child_env = dict(os.environ if env is None else env)
    for v in RENDEZVOUS_VARS + ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        child_env.pop(v, None)              # the launcher sets its own
    # the host driver supports dmabuf IPC only: RCCL and the direct-peer exchange both need this
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = rank_launch_command(nproc, script, script_args, port)
    sys.stdout.flush()
    sys.stderr.flush()
    return subprocess.run(cmd, env=child_env).returncode


class EngineFollowers:
    """Ranks 1 .. world-1 of an engine whose rank 0 lives in THIS process (see follower.py for the protocol).  Started
    before this process touches the GPU; afterwards this process is rank 0 of the same rendezvous."""

    def __init__(self, world: int, engine_kwargs: dict, env=None, port: Optional[int] = None, python: Optional[str] = None):
        import json

        self._json = json
        self.world = world
        port = port or free_port()
        base = dict(os.environ if env is None else env)
        base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        base.update(WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        base["PYTHONPATH"] = root + (os.pathsep + base["PYTHONPATH"] if base.get("PYTHONPATH") else "")
        self.procs = []
        for rank in range(1, world):
            e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
            self.procs.append(subprocess.Popen([python or sys.executable, "-m", "nano_vllm_x_amd.follower"], env=e, stdin=subprocess.PIPE,
                                               text=True, bufsize=1))
        # this process becomes rank 0 of the same job
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          LOCAL_WORLD_SIZE=str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.send({"op": "init", "kwargs": engine_kwargs})

    def send(self, event: dict):
        line = self._json.dumps(event) + "\n"
        for rank, p in enumerate(self.procs, start=1):
            if p.poll() is not None:
                raise RuntimeError(f"rank {rank} of the engine has exited with code {p.returncode}")
            try:
                p.stdin.write(line)
                p.stdin.flush()
            except (BrokenPipeError, OSError) as e:
                raise RuntimeError(f"rank {rank} of the engine is gone ({e})") from None

    def close(self, timeout: float = 60.0):
        for p in self.procs:
            try:
                if p.poll() is None:
                    p.stdin.write(self._json.dumps({"op": "shutdown"}) + "\n")
                    p.stdin.flush()
                p.stdin.close()
            except (BrokenPipeError, OSError, ValueError):
                pass
        for p in self.procs:
            try:
                p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
        self.procs = []


def require_launcher(world: int, what: str = "LLM"):
    """Fail at once, with the remedy, when a job of several ranks is asked for from a process nobody launched as a
    rank: joining the process group would otherwise wait for peers that were never started."""
    if world > 1 and not launcher_env_present():
        missing = [v for v in RENDEZVOUS_VARS if v not in os.environ]
        raise RuntimeError(
            f"{what} with {world} ranks must run as one process per GPU, but this process was not started as a rank "
            f"({', '.join(missing)} not set). Start the script with "
            f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {world} --master-addr 127.0.0.1 "
            f"--master-port <port> <script>` or call nano_vllm_x_amd.launch.run_ranks({world}, <script>) from a process "
            "that has not used the GPU; every rank then constructs the same LLM and submits the same requests.")
