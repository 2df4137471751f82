#!/usr/bin/env python3
"""Device idle time between decode steps, from a rocprofv3 --kernel-trace csv.

  python tools/step_gaps.py <dir with *_kernel_trace.csv>

A decode step starts with decode_step_setup_kernel.  Printed: per step, the gap between the end of the
kernel before it and its start (the host's critical path: ids to the host, scheduling, graph launch), the
step's span and the time inside the span in which no kernel ran (gaps between graph nodes)."""
import csv, glob, os, sys
import numpy as np

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("decode_step_setup_kernel")]
gaps, spans, inner = [], [], []
for a, b in zip(starts, starts[1:]):
    if a == 0:
        continue
    gaps.append(rows[a][0] - max(r[1] for r in rows[max(0, a - 4):a]))
    last = b - 1
    span = max(r[1] for r in rows[a:b]) - rows[a][0]
    busy = 0
    end = rows[a][0]
    for s, e, _ in rows[a:b]:
        if e > end:
            busy += e - max(s, end)
            end = e
    spans.append(span)
    inner.append(span - busy)
gaps, spans, inner = (np.array(x) / 1e3 for x in (gaps, spans, inner))
ok = gaps < 5000      # drop the prefill steps that sit between two decode steps
print(f"decode steps {len(gaps)}  gap before a step: mean {gaps[ok].mean():.1f} us, median {np.median(gaps[ok]):.1f}, p90 {np.percentile(gaps[ok], 90):.1f}")
print(f"step span (first kernel start to last kernel end, sampler included): mean {spans[ok].mean():.1f} us; idle inside the span: mean {inner[ok].mean():.1f} us")
print(f"sum over steps: gaps {gaps[ok].sum() / 1e3:.1f} ms, spans {spans[ok].sum() / 1e3:.1f} ms")
