"""Starting the ranks of a multi-GPU job from one command, and failing fast when nobody did.

The reference's LLM(tp_size=N) spawns its own worker processes (nanovllmx/core/executor.py:32-49,
core/worker_client.py:44-49).  Here the caller's script runs once per GPU; `launch.run_ranks` starts the copies
through torch.distributed.run as a fresh child job, `bench.py --gpus N` uses it when typed plainly, and a process
that asks for several ranks without having been started as one is told so at once.
"""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nano_vllm_x_amd import launch  # noqa: E402


def _clean_env():
    env = dict(os.environ)
    for v in launch.RENDEZVOUS_VARS + ("LOCAL_RANK",):
        env.pop(v, None)
    return env


def test_launch_command_is_one_torchrun_job_on_localhost():
    cmd = launch.rank_launch_command(4, "/x/bench.py", ["--gpus", "4", "--steps", "2"], port=29555, python="python3")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
                   "--master-port", "29555", "/x/bench.py", "--gpus", "4", "--steps", "2"]
    with pytest.raises(ValueError):
        launch.rank_launch_command(0, "x.py")
    # without a port one is picked that is free right now
    port = int(launch.rank_launch_command(2, "x.py")[8])
    assert 1024 < port < 65536


def test_launcher_env_detection():
    assert launch.launcher_env_present({"RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert not launch.launcher_env_present({"RANK": "0", "WORLD_SIZE": "2"})
    assert not launch.launcher_env_present({})


def test_run_ranks_starts_every_rank_and_returns_the_exit_code(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        print("rank", os.environ["RANK"], "of", os.environ["WORLD_SIZE"], "args", sys.argv[1:], flush=True)
        sys.exit(0 if os.environ["MASTER_ADDR"] == "127.0.0.1" else 5)
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from nano_vllm_x_amd.launch import run_ranks
        sys.exit(run_ranks(2, {str(script)!r}, ["--x", "1"]))
    """))
    r = subprocess.run([sys.executable, str(driver)], env=_clean_env(), capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    for k in (0, 1):
        assert f"rank {k} of 2 args ['--x', '1']" in r.stdout
    # a failing rank fails the job
    script.write_text("import sys; sys.exit(7)")
    r = subprocess.run([sys.executable, str(driver)], env=_clean_env(), capture_output=True, text=True, timeout=180)
    assert r.returncode != 0


def test_several_ranks_without_a_launcher_fail_at_once(monkeypatch):
    """With NVX_SPAWN_RANKS=0 (the caller starts its own ranks, or none may be started) a process that is no rank is told
    so at once; by default LLM starts the other ranks itself (next test)."""
    for v in launch.RENDEZVOUS_VARS:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("NVX_SPAWN_RANKS", "0")
    from nano_vllm_x_amd import LLM
    from nano_vllm_x_amd import distributed as tpd

    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="not started as a rank"):
        LLM(model="Qwen3-0.6B", tp_size=8)
    with pytest.raises(RuntimeError, match="torch.distributed.run"):
        LLM(model="Qwen3-0.6B", tp_size=2, pp_size=2)
    with pytest.raises(RuntimeError, match="RANK"):
        tpd.init_tensor_parallel(2)
    assert time.perf_counter() - t0 < 5.0       # seconds, not a rendezvous time-out
    launch.require_launcher(1)                  # one rank needs nobody


def test_bench_relaunches_itself_for_several_gpus(tmp_path, monkeypatch):
    """`python3 bench.py --gpus 2 ...` typed plainly hands the same arguments to a two-rank child job and returns its
    exit code, before anything touches a GPU (there is none here)."""
    calls = {}

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env

        class R:
            returncode = 11
        return R()

    for v in launch.RENDEZVOUS_VARS:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setattr(launch.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--num-seqs", "48"])
    import importlib

    bench = importlib.import_module("bench")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 11
    cmd = calls["cmd"]
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-8:] == ["--gpus", "2", "--steps", "1", "--warmup", "0", "--num-seqs", "48"]
    assert os.path.basename(cmd[-9]) == "bench.py"
    assert "WORLD_SIZE" not in calls["env"] and calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_engine_followers_replay_the_callers_events_in_order_on_every_rank(tmp_path, monkeypatch):
    """LLM(tp_size * pp_size > 1) from a plain process starts ranks 1.. itself and tells them what its caller does: the same
    constructor arguments, then every add / abort / step / reset in the order they happened (the reference spawns its
    workers and ships them every batch, core/executor.py:32-49,86-97).  Driven here with a stand-in engine, no device."""
    import dataclasses
    import json

    from nano_vllm_x_amd.core.common import SamplingParams

    log = tmp_path / "events"
    env = _clean_env()
    env.update(NVX_FOLLOWER_ENGINE="follower_standin:RecordingEngine", NVX_STANDIN_LOG=str(log),
               PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", ""))
    saved = {v: os.environ.get(v) for v in launch.RENDEZVOUS_VARS + ("LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    try:
        kwargs = dict(model="Qwen3-0.6B", tp_size=3, max_bs=8, device_ids=None, run_ahead=True)
        f = launch.EngineFollowers(3, kwargs, env=env)
        assert os.environ["RANK"] == "0" and os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"
        sp = SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=5)
        f.send({"op": "add", "id": "a", "ids": [1, 2, 3], "params": dataclasses.asdict(sp)})
        f.send({"op": "step"})
        f.send({"op": "add", "id": "b", "ids": [4], "params": dataclasses.asdict(sp)})
        f.send({"op": "step"})
        f.send({"op": "abort", "id": "a"})
        f.send({"op": "reset"})
        procs = list(f.procs)
        f.close(timeout=60)
        assert all(p.returncode == 0 for p in procs)
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
    want = ["init", "add", "step", "add", "step", "abort", "reset", "shutdown"]
    for rank in (1, 2):
        recs = [json.loads(line) for line in open(f"{log}.{rank}")]
        assert [r["call"] for r in recs] == want
        assert recs[0]["kwargs"] == kwargs and recs[0]["world"] == "3" and recs[0]["master"] == "127.0.0.1"
        assert recs[1] == {"call": "add", "id": "a", "ids": [1, 2, 3], "temperature": 0.6, "max_new_tokens": 5, "ignore_eos": True}
    # a rank that has died is reported at the next event, not by a hung collective
    bad_env = dict(env, NVX_FOLLOWER_ENGINE="follower_standin:Missing")
    try:
        g = launch.EngineFollowers(2, {}, env=bad_env)
        time.sleep(3.0)
        with pytest.raises(RuntimeError, match="rank 1 of the engine"):
            for _ in range(50):
                g.send({"op": "step"})
                time.sleep(0.1)
        g.close(timeout=5)
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
