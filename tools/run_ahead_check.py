#!/usr/bin/env python3
"""Token streams of the bench workload (Qwen3-0.6B geometry, sampling at temperature 0.6) with and without
run-ahead: the two runs must produce the same ids for every request."""
import asyncio, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nano_vllm_x_amd import LLM, SamplingParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prompts, max_tokens = bench.workload(n)


async def run(llm):
    async def one(i):
        sp = SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=max_tokens[i])
        return [o.token_id async for o in llm.generate(list(prompts[i]), sp, sequence_id=f"q{i}")]
    return await asyncio.gather(*(one(i) for i in range(n)))


digests = []
for ahead in (False, True):
    llm = LLM(model="Qwen3-0.6B", gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=400000, run_ahead=ahead)
    res = asyncio.run(run(llm))
    digests.append(hashlib.sha256(repr(res).encode()).hexdigest())
    print(f"run_ahead={ahead}: {sum(map(len, res))} tokens, {llm.engine.ahead_steps} steps queued ahead, digest {digests[-1][:16]}", flush=True)
    llm.shutdown()
    del llm
assert digests[0] == digests[1], "token streams differ"
print("identical")
