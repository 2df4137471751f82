#!/usr/bin/env python3
"""Shared-system-prompt workload (the reference ships none): one 512-token prefix shared by N requests,
suffix U[100,512] tokens, U[100,512] generated tokens each, seed 0; one warm-up request carries the
prefix into the radix tree first.  Run once reference-faithful (the tree is never read at admission)
and once with Scheduler(prefix_reuse=True); prints one JSON line per mode.

  python tools/prefix_bench.py [--model Qwen3-8B] [--num-seqs 256]
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch


def workload(n, seed=0, prefix_len=512):
    rnd = random.Random(seed)
    prefix = [rnd.randint(0, 10000) for _ in range(prefix_len)]
    reqs = []
    for _ in range(n):
        suffix = [rnd.randint(0, 10000) for _ in range(rnd.randint(100, 512))]
        reqs.append((prefix + suffix, rnd.randint(100, 512)))
    return prefix, reqs


async def run(llm, reqs):
    from nano_vllm_x_amd import SamplingParams

    async def one(i, p, n):
        c = 0
        async for _ in llm.generate(list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=n), sequence_id=f"q{i}"):
            c += 1
        return c

    return sum(await asyncio.gather(*(one(i, p, n) for i, (p, n) in enumerate(reqs))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen3-8B")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--kv-cache-size", type=int, default=600_000)
    a = ap.parse_args()
    from nano_vllm_x_amd import LLM

    prefix, reqs = workload(a.num_seqs)
    for reuse in (False, True):
        llm = LLM(model=a.model, gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=a.kv_cache_size,
                  prefix_reuse=reuse)
        asyncio.run(run(llm, [(prefix + [1, 2, 3], 8)]))       # warm-up: also what puts the prefix into the tree
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        generated = asyncio.run(run(llm, reqs))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        mgr = llm.engine.scheduler.kv_manager
        prompt_tokens = sum(len(p) for p, _ in reqs)
        print(json.dumps({"workload": f"{a.num_seqs} requests sharing a 512-token prefix, suffix U[100,512], gen U[100,512], seed 0",
                          "model": a.model, "prefix_reuse": reuse, "wall_s": round(wall, 3),
                          "generated_tokens_per_s": round(generated / wall, 1), "generated": generated,
                          "prompt_tokens": prompt_tokens, "prefix_hit_tokens": mgr.prefix_hit_tokens,
                          "hit_rate": round(mgr.prefix_hit_tokens / max(1, prompt_tokens), 4),
                          "engine_steps": dict(llm.engine.model_runner.stats)}), flush=True)
        llm.shutdown()
        del llm
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
