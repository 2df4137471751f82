#!/usr/bin/env python3
"""Phase timing inside the prefill attention tile loop, from the instrumented build of the library
(tools/build_variant.sh stamps prefill_attn "-DPF_STAMPS", copied over nano_vllm_x_amd/libnvx_hip.so on the box).

Uniform long sequences (steady state of the loop); prints, for waves 0 and 3 of two workgroups, the mean
shader-clock ticks spent between consecutive stamps of tiles 16..47 (blocks 8 and 9 of the launch = query blocks 119 and 118 of the first 8 192-token sequence, kv head 0):
  0 loop top | 1 next tile's copies issued | 2 QK^T MFMAs issued | 3 max, rescale done | 4 first probabilities
  | 5 PV done (issued) | 6 copies of the next tile landed | 7 barrier passed
"""
import ctypes
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from nano_vllm_x_amd import ops

DEV = "cuda:0"
HQ, HKV = 16, 8
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_seq = max(1, 131072 // L)
lens = [L] * n_seq
t_tok = sum(lens)
kc = torch.randn(2, t_tok + 1024, HKV, 128, device=DEV, dtype=torch.bfloat16)
vc = torch.randn_like(kc)
q = torch.randn(t_tok, HQ * 128, device=DEV, dtype=torch.bfloat16)
indptr = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).to(DEV)
idx = torch.randperm(t_tok + 1024, device=DEV)[:t_tok].to(torch.int32)
plan = ops.prefill_plan_ws(len(lens), DEV)
ops.prefill_plan(indptr, plan)
out = torch.empty_like(q)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(3):
    e0.record()
    ops.paged_prefill(q, kc[i % 2], vc[i % 2], indptr, indptr, idx, plan, HQ, HKV, 128 ** -0.5, out=out)
    e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = ctypes.CDLL(glob.glob(os.path.join(ROOT, "nano_vllm_x_amd", "libnvx_hip.so"))[0])
host = (ctypes.c_ulonglong * (2 * 2 * 32 * 8))()
rc = raw.nvx_debug_pf_stamps(host)
a = np.array(host[:], dtype=np.int64).reshape(2, 2, 32, 8)
flop = 4 * HQ * 128 * sum(l * (l + 1) // 2 for l in lens)
print(json.dumps({"rc": rc, "kernel_ms": round(ms, 3), "TFLOPs": round(flop / ms / 1e9, 1)}))
names = ["top->copies issued", "->QK issued", "->max/rescale", "->first probs", "->PV issued", "->copies landed", "->barrier", "->next top"]
for blk in range(2):
    for w in range(2):
        s = a[blk, w]
        if s.min() <= 0:
            print("block", blk, "wave", 3 * w, "no stamps")
            continue
        seg = np.diff(s, axis=1)                      # 7 segments inside a tile
        nxt = s[1:, 0] - s[:-1, 7]                    # barrier -> next loop top
        per_tile = s[1:, 0] - s[:-1, 0]
        print(f"block {blk} wave {3 * w}: ticks per tile {per_tile.mean():.0f} (min {per_tile.min()}, max {per_tile.max()})")
        print("   " + " | ".join(f"{n} {v:.0f}" for n, v in zip(names, list(seg.mean(0)) + [nxt.mean()])))
