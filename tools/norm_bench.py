import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from kernel_bench import timed
from nano_vllm_x_amd import ops
DEV="cuda:0"
for hidden in (2048, 4096, 5120):
    for rows in (1, 29, 123, 256):
        x = torch.randn(rows, hidden, device=DEV).to(torch.bfloat16); r = torch.randn_like(x); w = torch.ones(hidden, device=DEV, dtype=torch.bfloat16)
        t = timed(lambda: ops.rmsnorm(x, w, 1e-6, r), 28)
        g = torch.randn(rows, 2*3*hidden, device=DEV).to(torch.bfloat16)
        t2 = timed(lambda: ops.silu_mul(g), 28)
        print(hidden, rows, "rmsnorm us", round(t*1e6,2), "silu_mul us", round(t2*1e6,2))
