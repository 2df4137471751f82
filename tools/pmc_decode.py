#!/usr/bin/env python3
"""One launch of paged decode attention (the variant fed with the raw projection) per sampled decode batch of the bench workload, plus a
calibration copy of known size, for `rocprofv3 --pmc <counter> -- python3 tools/pmc_decode.py`.

Prints the algorithmic bytes of the sampled launches so that tools/pmc_summarise.py can set the
counters against them.
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from nano_vllm_x_amd import ops

DEV = "cuda:0"
HQ, HKV, LAYERS = 16, 8, 28
EVERY = int(sys.argv[1]) if len(sys.argv) > 1 else 8
OUT = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
if len(sys.argv) > 3:      # query,kv head counts of another geometry, e.g. 32,8 (Qwen3-8B on one GPU)
    HQ, HKV = (int(x) for x in sys.argv[3].split(","))


def main():
    prompts, max_tokens = bench.workload(256)
    slots = 200_000
    batches, n_dec, kv_rows = bench.decode_batches_of_workload(prompts, max_tokens, slots, 256, 4096, EVERY)
    kc = torch.randn(LAYERS, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    vc = torch.randn(LAYERS, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
    ws = ops.DecodeWorkspace(256, HQ, HKV, DEV)
    from nano_vllm_x_amd.layers.rotary_embedding import build_cos_sin_cache
    table = build_cos_sin_cache(128, 4096, 1e6).to(DEV)
    norm_w = torch.ones(128, device=DEV, dtype=torch.bfloat16)
    # calibration: a 1 GiB device-to-device copy (reads 1 GiB, writes 1 GiB) through a wide elementwise kernel
    cal = torch.empty(1 << 28, dtype=torch.float32, device=DEV).normal_()
    cal2 = torch.empty_like(cal)
    torch.cuda.synchronize()
    cal2.copy_(cal * 1.0001)     # elementwise kernel: one read, one write of 1 GiB each (plus the copy)
    torch.cuda.synchronize()
    alg = []
    layer = 0
    for lens, flat in batches:
        b = len(lens)
        kv_len = torch.from_numpy(lens).to(DEV)
        kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
        idx = torch.from_numpy(flat).to(DEV)
        qkv = torch.randn(b, (HQ + 2 * HKV) * 128, device=DEV, dtype=torch.bfloat16)
        out = torch.empty(b, HQ * 128, device=DEV, dtype=torch.bfloat16)
        pos = (kv_len - 1).to(torch.int64)
        new_slots = idx[kv_start + kv_len.to(torch.int64) - 1].to(torch.int64)
        ops.decode_plan(kv_len, kv_start, ws, out_slots=new_slots)
        ops.paged_decode_qkv(qkv, pos, table, norm_w, norm_w, 1e-6, kc[layer], vc[layer], idx, b, ws, 128 ** -0.5, out=out)
        layer = (layer + 1) % LAYERS
        rows = int(lens.sum())
        alg.append(rows * 2 * HKV * 128 * 2 + b * (HQ + 2 * HKV) * 128 * 2 + b * HQ * 128 * 2 + 4 * rows)
    torch.cuda.synchronize()
    info = {"heads": [HQ, HKV], "launches": len(alg), "avg_algorithmic_bytes": sum(alg) / len(alg), "calibration_bytes_each_way": cal.numel() * 4}
    print(json.dumps(info))
    if OUT:
        with open(OUT, "w") as f:
            json.dump(info, f)


if __name__ == "__main__":
    main()
