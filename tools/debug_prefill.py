"""Prefill attention against a plain torch reference on the device, for batches with far more work units
than persistent workgroups; prints the worst error per sequence and per 64-query block."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_vllm_x_amd import ops

DEV = "cuda:0"
hq, hkv = 16, 8
gen = torch.Generator().manual_seed(1)
lens = [int(x) for x in torch.randint(100, 1024, (int(sys.argv[1]) if len(sys.argv) > 1 else 96,), generator=gen)]
t = sum(lens)
q = torch.randn(t, hq, 128, generator=gen).to(torch.bfloat16).to(DEV)
kc = torch.randn(t, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
vc = torch.randn(t, hkv, 128, generator=gen).to(torch.bfloat16).to(DEV)
indptr_np = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
indptr = torch.from_numpy(indptr_np).to(DEV)
idx = torch.arange(t, dtype=torch.int32, device=DEV)
plan = ops.prefill_plan_ws(len(lens), DEV)
ops.prefill_plan(indptr, plan)
out = ops.paged_prefill(q.view(t, hq * 128), kc, vc, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5).view(t, hq, 128).float()
torch.cuda.synchronize()
bad = 0
for b, L in enumerate(lens):
    s = int(indptr_np[b])
    qq = q[s:s + L].float().permute(1, 0, 2)                       # (hq, L, d)
    kk = kc[s:s + L].float().permute(1, 0, 2).repeat_interleave(hq // hkv, 0)
    vv = vc[s:s + L].float().permute(1, 0, 2).repeat_interleave(hq // hkv, 0)
    sc = qq @ kk.transpose(1, 2) * 128 ** -0.5
    sc = sc.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool, device=DEV), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vv).permute(1, 0, 2)
    err = (out[s:s + L] - ref).abs().amax(dim=(1, 2))              # per query
    if err.max() > 0.05:
        bad += 1
        blocks = [(i // 64, round(float(err[i:i + 64].max()), 3)) for i in range(0, L, 64) if err[i:i + 64].max() > 0.05]
        heads = (out[s:s + L] - ref).abs().amax(dim=(0, 2))
        print(f"seq {b} len {L}: bad blocks {blocks[:8]} heads {[i for i in range(hq) if heads[i] > 0.05]}")
print("sequences with errors:", bad, "of", len(lens))
out2 = ops.paged_prefill(q.view(t, hq * 128), kc, vc, indptr, indptr, idx, plan, hq, hkv, 128 ** -0.5).view(t, hq, 128).float()
torch.cuda.synchronize()
d = (out2 - out).abs().amax(dim=(1, 2))
rows = torch.nonzero(d > 0).flatten().tolist()
print("rows that differ between two runs:", len(rows), rows[:10])
for r in rows[:6]:
    b = int(np.searchsorted(indptr_np, r, side="right") - 1)
    print("  row", r, "seq", b, "pos", r - int(indptr_np[b]), "len", lens[b], "diff", float(d[r]))
