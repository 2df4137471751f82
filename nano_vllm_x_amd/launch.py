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
