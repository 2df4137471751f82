#!/bin/bash
# Final artefacts of a round: the default bench line, then the same workload under
# `rocprofv3 --kernel-trace --stats` (kernel summary as csv).  Everything lands under gpurun_out/.
set -e
mkdir -p gpurun_out/tmp
export TMPDIR=$PWD/gpurun_out/tmp
timeout -k 10 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 400 gpurun_out/bench_final.json; echo
rm -rf gpurun_out/prof_final
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench_final.json 2> gpurun_out/prof_bench_final.err
f=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/kernel_stats_final.csv
rm -rf gpurun_out/prof_final gpurun_out/tmp
head -8 gpurun_out/kernel_stats_final.csv | cut -c1-160
