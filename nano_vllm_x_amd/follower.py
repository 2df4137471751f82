"""Rank >= 1 of an engine that `LLM(tp_size * pp_size > 1)` started from a plain process.

The reference's LLM spawns one worker process per GPU and ships every step's batch to each of them (nanovllmx/
core/executor.py:32-49,86-97, core/worker_client.py:44-49,87-119).  Here every rank runs the same deterministic engine, so
what the ranks started by `launch.EngineFollowers` need from rank 0 is only what rank 0's CALLER did, in order: one line of
JSON per event on their standard input --

    {"op": "init", "kwargs": {...}}            the Engine's constructor arguments (first line)
    {"op": "add", "id": ..., "ids": [...], "params": {...}}
    {"op": "abort", "id": ...}
    {"op": "step"}                             one Engine.step_overlapped()
    {"op": "reset"} / {"op": "shutdown"}

-- a few dozen bytes per step instead of a pickled batch; the sampled ids reach every rank through the collectives of the
step itself.  The process ends with its parent (PR_SET_PDEATHSIG, as utils.py:10-17 does for the reference's workers) or at
end of input.

NVX_FOLLOWER_ENGINE=module:callable replaces the Engine class (tests drive the protocol with a stand-in on CPU).
"""
from __future__ import annotations

import importlib
import json
import os
import signal
import sys


def die_with_parent():
    try:
        import ctypes

        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG
    except OSError:
        pass


def engine_factory():
    spec = os.environ.get("NVX_FOLLOWER_ENGINE")
    if spec:
        mod, name = spec.split(":")
        return getattr(importlib.import_module(mod), name)
    from .core.engine import Engine

    return Engine


def serve(lines, factory=None) -> int:
    """Apply the leader's events to this rank's engine until "shutdown" or end of input."""
    from .core.common import SamplingParams

    factory = factory or engine_factory()
    eng = None
    for line in lines:
        line = line.strip()
        if not line:
            continue
        ev = json.loads(line)
        op = ev["op"]
        if op == "init":
            eng = factory(**ev["kwargs"])
        elif op == "add":
            eng.add_sequence(sequence_id=ev["id"], prompt_token_ids=ev["ids"], sampling_params=SamplingParams(**ev["params"]))
        elif op == "abort":
            eng.abort_sequence(ev["id"])
        elif op == "step":
            eng.step_overlapped()
        elif op == "reset":
            eng.reset()
        elif op == "shutdown":
            break
        else:
            raise ValueError(f"unknown event {op!r}")
    if eng is not None:
        eng.shutdown()
    return 0


def main() -> int:
    die_with_parent()
    return serve(sys.stdin)


if __name__ == "__main__":
    sys.exit(main())
