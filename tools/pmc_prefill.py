#!/usr/bin/env python3
"""A few launches of paged prefill attention on the bench workload's 256 prompts, for
`rocprofv3 --pmc <SQ counters> -- python3 tools/pmc_prefill.py` (MFMA busy cycles, LDS conflicts)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nano_vllm_x_amd import ops

DEV = "cuda:0"
HQ, HKV = 16, 8
prompts, _ = bench.workload(256)
lens = [len(p) for p in prompts]
t_tok = sum(lens)
slots = t_tok + 1024
kc = torch.randn(2, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
vc = torch.randn(2, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
q = torch.randn(t_tok, HQ * 128, device=DEV, dtype=torch.bfloat16)
indptr = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).to(DEV)
idx = torch.arange(t_tok, dtype=torch.int32, device=DEV)
plan = ops.prefill_plan_ws(len(lens), DEV)
ops.prefill_plan(indptr, plan)
out = torch.empty_like(q)
for i in range(4):
    ops.paged_prefill(q, kc[i % 2], vc[i % 2], indptr, indptr, idx, plan, HQ, HKV, 128 ** -0.5, out=out)
torch.cuda.synchronize()
flop = 4 * HQ * 128 * sum(l * (l + 1) // 2 for l in lens)
# bytes every launch must move at least once: q and out rows (HQ x 256 B per token each), K and V rows (HKV x 256 B per token each)
compulsory = t_tok * (2 * HQ * 256 + 2 * HKV * 256) + 4 * t_tok
info = {"launches": 4, "algorithmic_flop_per_launch": flop, "tokens": t_tok, "compulsory_bytes_per_launch": compulsory}
print(json.dumps(info))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(info, f)
