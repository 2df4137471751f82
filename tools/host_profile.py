#!/usr/bin/env python3
"""cProfile of one pass of the bench workload: where the host spends its time between GPU steps."""
import asyncio, cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nano_vllm_x_amd import LLM

prompts, max_tokens = bench.workload(256)
llm = LLM(model="Qwen3-0.6B", gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=400000)
asyncio.run(bench.one_pass(llm, prompts, max_tokens))
llm.engine.reset()
pr = cProfile.Profile()
pr.enable()
asyncio.run(bench.one_pass(llm, prompts, max_tokens))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
