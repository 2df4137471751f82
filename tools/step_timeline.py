#!/usr/bin/env python3
"""Where a decode step's time goes, kernel by kernel, from a rocprofv3 --kernel-trace csv.

  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/step_timeline.py run [--batch 29] [--prompt 560]
  python3 tools/step_timeline.py analyze DIR

`run` drives the engine through one prefill and a few dozen hipGraph decode steps at a fixed batch (requests of
equal length, ignore_eos).  `analyze` takes the decode steps of the trace (a step starts with
decode_step_setup_kernel), and prints, per kernel of a step in launch order, its mean duration and the mean idle
time between the end of the kernel before it and its start, averaged over steps and -- for the per-layer
kernels -- over layers."""
from __future__ import annotations

import argparse
import asyncio
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import random

    from nano_vllm_x_amd import LLM, SamplingParams

    llm = LLM(model=args.model, gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=400_000, run_ahead=True)
    rnd = random.Random(0)
    prompts = [[rnd.randint(0, 10000) for _ in range(args.prompt)] for _ in range(args.batch)]

    async def go():
        async def one(i, p):
            async for _ in llm.generate(list(p), SamplingParams(temperature=0.6, ignore_eos=True, max_new_tokens=args.steps), sequence_id=f"t{i}"):
                pass
        await asyncio.gather(*(one(i, p) for i, p in enumerate(prompts)))

    asyncio.run(go())
    llm.engine.reset()
    asyncio.run(go())
    print(dict(llm.engine.model_runner.stats))
    llm.shutdown()


def short(name: str) -> str:
    name = name.replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:60]


def analyze(args):
    rows = []
    for f in glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("decode_step_setup_kernel")]
    steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
    steps = steps[len(steps) // 2:]          # the second pass of `run`: warmed up
    if not steps:
        print("no decode steps in the trace")
        return
    n_kernels = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
    steps = [s for s in steps if len(s) == n_kernels]
    dur = defaultdict(list)
    gap = defaultdict(list)
    order = []
    spans = []
    for s in steps:
        spans.append(s[-1][1] - s[0][0])
        seen = defaultdict(int)
        for i, (st, en, name) in enumerate(s):
            k = short(name)
            seen[k] += 1
            if k not in order:
                order.append(k)
            dur[k].append(en - st)
            if i:
                gap[k].append(st - s[i - 1][1])
    per_step = {k: len(dur[k]) / len(steps) for k in order}
    print(f"{len(steps)} decode steps of {n_kernels} kernels; mean span {sum(spans) / len(spans) / 1e3:.1f} us")
    print(f"{'kernel':60s} {'per step':>8s} {'mean us':>8s} {'gap before us':>13s} {'us per step':>11s}")
    tot = totgap = 0.0
    for k in order:
        d = sum(dur[k]) / len(dur[k]) / 1e3
        g = sum(gap[k]) / max(1, len(gap[k])) / 1e3
        tot += d * per_step[k]
        totgap += g * per_step[k]
        print(f"{k:60s} {per_step[k]:8.1f} {d:8.2f} {g:13.2f} {(d + g) * per_step[k]:11.1f}")
    print(f"kernels {tot:.1f} us + gaps {totgap:.1f} us per step")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["run", "analyze"])
    ap.add_argument("dir", nargs="?", default="")
    ap.add_argument("--batch", type=int, default=29)
    ap.add_argument("--prompt", type=int, default=560)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--model", default="Qwen3-0.6B")
    a = ap.parse_args()
    run(a) if a.cmd == "run" else analyze(a)
