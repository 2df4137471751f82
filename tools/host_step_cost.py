#!/usr/bin/env python3
"""Host cost of the two calls that stand between two device steps (Scheduler.schedule and
EngineCore.fold_output) over the bench workload, without a device: the sampled ids are a constant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from nano_vllm_x_amd.core.common import SamplingParams
from nano_vllm_x_amd.core.engine import EngineCore

p, m = bench.workload(256)
core = EngineCore(kv_cache_size=2_000_000, max_bs=256, context_len=4096)
for i, (pp, mm) in enumerate(zip(p, m)):
    core.add_sequence(f"s{i}", list(pp), SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=mm))
ts, tu, bs = [], [], []
while core.has_unfinished_sequences():
    t0 = time.perf_counter(); b = core.scheduler.schedule(); t1 = time.perf_counter()
    if not b:
        break
    ids = [7] * b.num_seqs
    t2 = time.perf_counter(); core.fold_output(b, ids); t3 = time.perf_counter()
    ts.append(t1 - t0); tu.append(t3 - t2); bs.append(b.num_seqs)
ts, tu, bs = map(np.array, (ts, tu, bs))
print(f"steps {len(ts)}  schedule {ts[2:].mean() * 1e6:.1f} us  fold {tu[2:].mean() * 1e6:.1f} us  per pass {ts.sum() + tu.sum():.3f} s")
for lo, hi in ((1, 16), (17, 48), (49, 256)):
    k = (bs >= lo) & (bs <= hi)
    print(f"  batch {lo}-{hi}: {k.sum()} steps, schedule {ts[k].mean() * 1e6:.1f} us, fold {tu[k].mean() * 1e6:.1f} us")
