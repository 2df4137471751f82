"""How much do the library's decode-sized projections of Qwen3-8B gain from PyTorch's TunableOp (a search over the
BLAS libraries' own solutions, done once per shape)?  Prints us per call with the default heuristic and after tuning.

    python tools/gemm_tune_probe.py [--rows 8,32,64,128,256] [--file gpurun_out/tunableop_8b.csv]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (24576, 4096), "down": (4096, 12288)}


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="8,32,64,128,256")
    ap.add_argument("--file", default="gpurun_out/tunableop_8b.csv")
    ap.add_argument("--copies", type=int, default=6, help="weight copies cycled through, so that no call finds its weight in a cache")
    args = ap.parse_args()
    dev = "cuda:0"
    rows = [int(r) for r in args.rows.split(",")]
    import torch.cuda.tunable as tn
    for name, (n, k) in SHAPES.items():
        ws = [(torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16) for _ in range(args.copies)]
        for m in rows:
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            i = [0]

            def run():
                i[0] = (i[0] + 1) % args.copies
                return F.linear(x, ws[i[0]])

            tn.enable(False)
            base = timed(run, 30)
            tn.enable(True)
            tn.tuning_enable(True)
            tn.set_max_tuning_duration(30)
            tn.set_max_tuning_iterations(10)
            tn.set_filename(args.file)
            t0 = time.time()
            run()
            torch.cuda.synchronize()
            tune_s = time.time() - t0
            tn.tuning_enable(False)
            tuned = timed(run, 30)
            tn.enable(False)
            print(json.dumps({"layer": name, "M": m, "N": n, "K": k, "default_us": round(base, 1), "tuned_us": round(tuned, 1),
                              "tuning_s": round(tune_s, 1), "weight_GBps_tuned": round(n * k * 2 / tuned / 1e3)}), flush=True)
        del ws
    tn.write_file(args.file)


if __name__ == "__main__":
    sys.exit(main())
