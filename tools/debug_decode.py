#!/usr/bin/env python3
"""Tiny decode-attention cases with hand-checkable answers (debugging aid, run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nano_vllm_x_amd import ops
DEV = "cuda:0"

def case(lens, hq, hkv, slots=64):
    b = len(lens)
    kc = torch.zeros(slots, hkv, 128, dtype=torch.bfloat16)
    vc = torch.zeros(slots, hkv, 128, dtype=torch.bfloat16)
    for s in range(slots):
        for h in range(hkv):
            vc[s, h] = s + 0.25 * h + torch.arange(128) / 1024.0
    q = torch.zeros(b, hq * 128, dtype=torch.bfloat16)
    total = sum(lens)
    idx = torch.arange(total, dtype=torch.int32) + 3
    start = torch.tensor([sum(lens[:i]) for i in range(b)], dtype=torch.int64)
    ws = ops.DecodeWorkspace(b, hq, hkv, DEV)
    ops.decode_plan(torch.tensor(lens, dtype=torch.int32, device=DEV), start.to(DEV), ws)
    out = ops.paged_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), idx.to(DEV), b, ws, 0.1)
    torch.cuda.synchronize()
    print("lens", lens, "hq", hq, "hkv", hkv, "plan", ws.plan[:24].tolist())
    o = out.view(b, hq, 128).float().cpu()
    for i in range(b):
        sl = idx[int(start[i]):int(start[i]) + lens[i]].float()
        for h in range(hq):
            want = sl.mean() + 0.25 * (h // (hq // hkv))
            print("  seq", i, "head", h, "want[0]~", float(want), "got", o[i, h, :4].tolist(), o[i, h, 124:].tolist())

case([1], 1, 1)
case([5], 1, 1)
case([5], 2, 1)
case([40, 3], 2, 1)
case([200], 4, 2)

def dot_case(n, hq, hkv, dims):
    slots = 512
    g = torch.Generator().manual_seed(1)
    kc = torch.randn(slots, hkv, 128, generator=g).to(torch.bfloat16)
    vc = torch.randn(slots, hkv, 128, generator=g).to(torch.bfloat16)
    q = torch.zeros(1, hq, 128)
    q[..., dims] = torch.randn(1, hq, len(dims), generator=g) * 3
    q = q.to(torch.bfloat16)
    idx = torch.arange(n, dtype=torch.int32) + 3
    ws = ops.DecodeWorkspace(1, hq, hkv, DEV)
    ops.decode_plan(torch.tensor([n], dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV), ws)
    out = ops.paged_decode(q.view(1, -1).to(DEV), kc.to(DEV), vc.to(DEV), idx.to(DEV), 1, ws, 0.1)
    o = out.view(hq, 128).float().cpu()
    G = hq // hkv
    for h in range(hq):
        k = kc[3:3 + n, h // G].float(); v = vc[3:3 + n, h // G].float()
        p = torch.softmax((k @ q[0, h].float()) * 0.1, 0)
        want = p @ v
        print("  n", n, "hq/hkv", hq, hkv, "dims", dims[:4], "head", h, "max err", float((o[h] - want).abs().max()))

for dims in ([0], [1], [8], [64], [127], list(range(128))):
    dot_case(1, 1, 1, dims); dot_case(2, 1, 1, dims); dot_case(9, 2, 1, dims)
