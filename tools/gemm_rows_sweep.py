"""Library projection time against the number of rows, for the decode-step projections of a geometry: does the BLAS
heuristic ever pick a kernel for M rows that is slower than the one it picks for more rows (so that padding would pay)?

    python tools/gemm_rows_sweep.py [--hidden 4096 --inter 12288 --hq 32 --hkv 8]
"""
import argparse
import json

import torch
import torch.nn.functional as F

DEV = "cuda:0"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=12288)
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    args = ap.parse_args()
    h = args.hidden
    shapes = {"qkv": ((args.hq + 2 * args.hkv) * 128, h), "o": (h, args.hq * 128), "gate_up": (2 * args.inter, h), "down": (h, args.inter)}
    rows = [1, 2, 4, 8] + list(range(16, 257, 16))
    for name, (n, k) in shapes.items():
        copies = max(3, -(-(768 << 20) // (n * k * 2)))     # more than the memory-side cache holds
        ws = [(torch.randn(n, k, device=DEV) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
        xs = torch.randn(256, k, device=DEV).to(torch.bfloat16)
        out = {}
        for m in rows:
            x = xs[:m]
            i = [0]

            def run():
                i[0] = (i[0] + 1) % copies
                F.linear(x, ws[i[0]])

            out[m] = round(timed(run), 1)
        best_from = {}
        for m in rows:      # cheapest size >= m
            cand = min((out[m2], m2) for m2 in rows if m2 >= m)
            if cand[1] != m and cand[0] < 0.93 * out[m]:
                best_from[m] = {"pad_to": cand[1], "us": cand[0], "instead_of": out[m]}
        print(json.dumps({"layer": name, "N": n, "K": k, "us_by_rows": out, "padding_pays": best_from}), flush=True)
        del ws


if __name__ == "__main__":
    main()
