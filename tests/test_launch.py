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
    for v in launch.RENDEZVOUS_VARS:
        monkeypatch.delenv(v, raising=False)
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
