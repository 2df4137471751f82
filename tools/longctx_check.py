#!/usr/bin/env python3
"""Long-context serving check on one GPU: context 4096, max_bs 256, prompts
U[1024,3072], U[64,256] new tokens; a KV pool small enough that admission has to wait for running
sequences.  Prints one JSON line; fails loudly if any request ends with the wrong token count."""
import asyncio, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nano_vllm_x_amd import LLM, SamplingParams

model = sys.argv[1] if len(sys.argv) > 1 else "Qwen3-0.6B"
n_req = int(sys.argv[2]) if len(sys.argv) > 2 else 192
rnd = random.Random(0)
reqs = [([rnd.randint(0, 10000) for _ in range(rnd.randint(1024, 3072))], rnd.randint(64, 256)) for _ in range(n_req)]
pool = 150_000        # < sum of final lengths (~430k): several admission waves
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # > 0: chunked prefill with this many tokens per prefill step
ahead = len(sys.argv) > 4 and sys.argv[4] == "ahead"         # queue decode steps ahead of their input ids
llm = LLM(model=model, gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=pool,
          max_prefill_tokens=chunk or None, chunked_prefill=chunk > 0, run_ahead=ahead)


async def run():
    async def one(i, p, n):
        c = 0
        async for o in llm.generate(list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=n), sequence_id=f"l{i}"):
            c += 1
        assert c == n, (i, c, n)
        return c
    return sum(await asyncio.gather(*(one(i, p, n) for i, (p, n) in enumerate(reqs))))

t0 = time.perf_counter()
gen = asyncio.run(run())
torch.cuda.synchronize()
wall = time.perf_counter() - t0
st = llm.engine.model_runner.stats
print(json.dumps({"model": model, "requests": n_req, "prompt_tokens": sum(len(p) for p, _ in reqs), "generated": gen,
                  "wall_s": round(wall, 2), "generated_tok_per_s": round(gen / wall, 1), "kv_pool_slots": pool, "prefill_chunk_tokens": chunk, "steps_queued_ahead": llm.engine.ahead_steps,
                  "engine_steps": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}}))
llm.shutdown()
