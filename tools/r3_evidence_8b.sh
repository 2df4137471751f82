#!/bin/bash
# BASELINE configuration 3 at its real geometry (Qwen3-8B, one GPU): the bench line, the kernel summary of the same
# command under rocprofv3, and the shared-prefix workload with prefix reuse off and on.  Everything under gpurun_out/.
set -e
mkdir -p gpurun_out/tmp
export TMPDIR=$PWD/gpurun_out/tmp
timeout -k 10 400 python bench.py --model Qwen3-8B --no-cpu-baseline > gpurun_out/r3_bench_qwen3_8b.json 2> gpurun_out/r3_bench_qwen3_8b.err
tail -c 600 gpurun_out/r3_bench_qwen3_8b.json; echo
rm -rf gpurun_out/prof_8b
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_8b -- python3 bench.py --model Qwen3-8B --steps 1 --warmup 1 --no-cpu-baseline --plain-pass 0 > gpurun_out/r3_bench_qwen3_8b_under_rocprof.json 2> gpurun_out/r3_bench_qwen3_8b_under_rocprof.err
f=$(find gpurun_out/prof_8b -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r3_bench_kernel_stats_qwen3_8b.csv
rm -rf gpurun_out/prof_8b
head -6 gpurun_out/r3_bench_kernel_stats_qwen3_8b.csv | cut -c1-150
timeout -k 10 400 python tools/prefix_bench.py --model Qwen3-8B > gpurun_out/r3_prefix_reuse_config3.jsonl 2> gpurun_out/r3_prefix_8b.err
timeout -k 10 300 python tools/prefix_bench.py --model Qwen3-0.6B >> gpurun_out/r3_prefix_reuse_config3.jsonl 2> gpurun_out/r3_prefix_06b.err
cat gpurun_out/r3_prefix_reuse_config3.jsonl | cut -c1-400
rm -rf gpurun_out/tmp
