"""LLM(tp_size=2) typed in a plain process: it starts rank 1 itself (launch.EngineFollowers), both ranks share the one GPU of
the test box and talk through gloo, and the caller sees the reference's API unchanged (nanovllmx/llm.py:17-39 reaches
core/executor.py:32-49 the same way).  The hidden size is 1024 so that the decode steps end in the one-pass LM head with the
ranks exchanging one key per row."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import asyncio, json, os, random, sys
sys.path.insert(0, %r)
os.environ["NVX_TP_BACKEND"] = "gloo"
os.environ["NVX_EXTRA_PRESETS"] = json.dumps({"spawned": dict(vocab_size=2048, hidden_size=1024, intermediate_size=1536, num_hidden_layers=2,
    num_attention_heads=8, num_key_value_heads=4, max_position_embeddings=512, tie_word_embeddings=False)})
from nano_vllm_x_amd import LLM, SamplingParams

llm = LLM(model="spawned", max_bs=8, tp_size=2, context_len=128, kv_cache_size=2048, run_ahead=True)
assert llm.followers is not None and len(llm.followers.procs) == 1
rnd = random.Random(0)
reqs = [([rnd.randint(0, 2000) for _ in range(rnd.randint(5, 40))], rnd.randint(3, 12)) for _ in range(6)]

async def main():
    async def one(i, p, n):
        return [o.token_id async for o in llm.generate(list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=n), sequence_id=f"r{i}")]
    return await asyncio.gather(*(one(i, p, n) for i, (p, n) in enumerate(reqs)))

first = asyncio.run(main())
llm.reset()
second = asyncio.run(main())          # after a reset on every rank: the engine serves again
stats = dict(llm.engine.model_runner.stats)
procs = list(llm.followers.procs)
llm.shutdown()
print("RESULT " + json.dumps({"want": [n for _, n in reqs], "first": first, "second": second, "stats": stats,
                              "follower_exit_codes": [p.returncode for p in procs]}))
""" % ROOT


def test_llm_with_two_ranks_from_a_plain_process():
    # a plain process: no launcher variables, and none of the switches other tests of this session may have left set
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK") and not k.startswith("NVX_")}
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=400)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")), None)
    assert r.returncode == 0 and line, r.stdout[-1500:] + r.stderr[-3000:]
    res = json.loads(line[len("RESULT "):])
    for got in (res["first"], res["second"]):
        assert [len(t) for t in got] == res["want"]
        assert all(0 <= tok < 2048 for t in got for tok in t)
    assert res["follower_exit_codes"] == [0]
    st = res["stats"]
    assert st["decode_steps"] > 0 and st["graph_steps"] == st["decode_steps"], st      # captured decode steps on both ranks
    assert st.get("fused_lm_head_steps", 0) > 0, st
