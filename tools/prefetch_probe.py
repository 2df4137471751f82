"""Does a weight matrix read by the kernel BEFORE a skinny projection make that projection faster
(lines left in the Infinity Cache / L2 across the kernel boundary)?  Times graph replays of
[touch(W[j]); linear(x, W[i])] with j == i (warm) and j == i + n/2 (cold) over many distinct matrices."""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_vllm_x_amd import ops  # noqa: E402

DEV = "cuda:0"


def replay_time(build, reps=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
        build()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3   # us


@torch.inference_mode()
def main():
    n_mats = 64
    for (n, k, mode, name) in [(4096, 1024, 0, "qkv"), (6144, 1024, 1, "gate_up"), (1024, 2048, 2, "o"), (1024, 3072, 2, "down")]:
        ws = [torch.randn(n, k, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(n_mats)]
        for m in (8, 29, 64):
            x = torch.randn(m, k, device=DEV, dtype=torch.bfloat16)
            sink = torch.zeros(n_mats, device=DEV, dtype=torch.float32)
            splits = ops.linear_decode_splits(n, k) if mode == 2 else 1

            def lin(i):
                ops.linear_decode(x, ws[i], mode=mode, k_splits=splits)

            def touch(j):
                sink[j] = ws[j].view(torch.int32).view(-1).sum(dtype=torch.int64).float()

            lin(0); touch(0)
            torch.cuda.synchronize()
            t_lin = replay_time(lambda: [lin(i) for i in range(n_mats)])
            t_touch = replay_time(lambda: [touch(i) for i in range(n_mats)])
            t_cold = replay_time(lambda: [(touch((i + n_mats // 2) % n_mats), lin(i)) for i in range(n_mats)])
            t_warm = replay_time(lambda: [(touch(i), lin(i)) for i in range(n_mats)])
            print(f"{name:8s} M={m:3d}  linear alone {t_lin / n_mats:6.2f}  touch alone {t_touch / n_mats:6.2f}  "
                  f"touch+linear cold {t_cold / n_mats:6.2f}  warm {t_warm / n_mats:6.2f}  saved {(t_cold - t_warm) / n_mats:5.2f} us", flush=True)
        del ws


if __name__ == "__main__":
    main()
