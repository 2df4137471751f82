#!/usr/bin/env python3
"""How long the device waits for the host between two steps of the bench workload: an event pair around
every step's device work (recorded on the stream, so it costs the device nothing) against the wall time
of the pass."""
import asyncio, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nano_vllm_x_amd import LLM
from nano_vllm_x_amd.core.common import ForwardMode

prompts, max_tokens = bench.workload(256)
ahead = len(sys.argv) > 1 and sys.argv[1] == "ahead"
llm = LLM(model="Qwen3-0.6B", gpu_memory_utilization=0.8, max_bs=256, context_len=4096, kv_cache_size=400000, run_ahead=ahead)
runner = llm.engine.model_runner
asyncio.run(bench.one_pass(llm, prompts, max_tokens))
llm.engine.reset()
events = []
real = runner.launch_step


def launch_step(batch, ids_on_device=False):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    h = real(batch, ids_on_device)
    b.record()
    events.append((a, b, batch.forward_mode, batch.num_seqs))
    return h


runner.launch_step = launch_step
torch.cuda.synchronize()
t0 = time.perf_counter()
asyncio.run(bench.one_pass(llm, prompts, max_tokens))
torch.cuda.synchronize()
wall = time.perf_counter() - t0
spans = [(a.elapsed_time(b), m, n) for a, b, m, n in events]
dec = [s for s, m, n in spans if m == ForwardMode.DECODE]
pre = [s for s, m, n in spans if m == ForwardMode.PREFILL]
between = [events[i][1].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]
print(f"wall {wall * 1e3:.1f} ms; device spans: prefill {sum(pre):.1f} ms ({len(pre)} steps), decode {sum(dec):.1f} ms ({len(dec)} steps, "
      f"mean {sum(dec) / len(dec) * 1e3:.1f} us)")
print(f"between the end of one step's device work and the start of the next: total {sum(between):.1f} ms, "
      f"mean {sum(between) / len(between) * 1e3:.1f} us, max {max(between) * 1e3:.1f} us")
for lo, hi in ((1, 16), (17, 48), (49, 256)):
    k = [i for i in range(len(events) - 1) if lo <= events[i + 1][3] <= hi and events[i + 1][2] == ForwardMode.DECODE]
    if k:
        print(f"  next batch {lo}-{hi}: {len(k)} steps, gap mean {sum(between[i] for i in k) / len(k) * 1e3:.1f} us, "
              f"step mean {sum(spans[i + 1][0] for i in k) / len(k) * 1e3:.1f} us")
