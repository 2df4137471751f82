#!/usr/bin/env python3
"""Phase timing inside the LDS-ring decode attention kernel, from the instrumented build of the library
(tools/build_variant.sh stamps decode_attn "-DDEC_STAMPS", copied over nano_vllm_x_amd/libnvx_hip.so on the box; run with
NVX_DEC_RING=2).  One launch over long sequences at a given batch and head geometry; prints the mean shader-clock ticks
per stage of 16 tokens between the stamps: loop top -> copies of the stage ahead issued -> this stage landed -> reduced."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HQ, HKV = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,8").split(","))
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
slots = B * L + 64
kc = torch.randn(4, slots, HKV, 128, device=DEV, dtype=torch.bfloat16)
vc = torch.randn_like(kc)
lens = np.full(B, L, dtype=np.int32)
kv_len = torch.from_numpy(lens).to(DEV)
kv_start = torch.from_numpy(np.concatenate(([0], np.cumsum(lens[:-1]))).astype(np.int64)).to(DEV)
idx = torch.randperm(slots, device=DEV)[: B * L].to(torch.int32)
q = torch.randn(B, HQ * 128, device=DEV, dtype=torch.bfloat16)
out = torch.empty_like(q)
ws = ops.DecodeWorkspace(max(B, 256), HQ, HKV, DEV)
ops.decode_plan(kv_len, kv_start, ws)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    e0.record()
    ops.paged_decode(q, kc[i], vc[i], idx, B, ws, 128 ** -0.5, out=out)
    e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
raw = ctypes.CDLL(glob.glob(os.path.join(ROOT, "nano_vllm_x_amd", "libnvx_hip.so"))[0])
host = (ctypes.c_ulonglong * (16 * 4 * 16 * 4))()
rc = raw.nvx_debug_dec_stamps(host)
a = np.array(host[:], dtype=np.int64).reshape(16 * 4, 16, 4)
nbytes = B * L * HKV * 512
print(json.dumps({"rc": rc, "B": B, "heads": [HQ, HKV], "L": L, "items": int(ws.plan[0]), "chunk": int(ws.plan[1]), "pair_us": round(us, 1),
                  "GBps": round(nbytes / us / 1e3, 1)}))
rows = []
for w in range(a.shape[0]):
    s = a[w]
    n = int((s[:, 0] > 0).sum())
    if n < 3:
        continue
    s = s[:n]
    seg = np.diff(s, axis=1)
    per = np.diff(s[:, 0])
    rows.append((per.mean(), seg[:, 0].mean(), seg[:, 1].mean(), seg[:, 2].mean(), (s[1:, 0] - s[:-1, 3]).mean(), n))
r = np.array(rows).reshape(-1, 6)
if not len(r):
    raise SystemExit("no wave with three stages")
print(f"{len(rows)} waves with >= 3 stages; mean ticks per stage {r[:, 0].mean():.0f}: issue {r[:, 1].mean():.0f} | wait {r[:, 2].mean():.0f} | reduce {r[:, 3].mean():.0f} | loop {r[:, 4].mean():.0f}; stages per wave {r[:, 5].mean():.1f}")
