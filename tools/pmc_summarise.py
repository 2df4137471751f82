#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for the decode-attention launches.

usage: pmc_summarise.py info.json fetch.csv [write.csv] > profiles/pmc_traffic.json

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte
request of a wide (16 B per lane) read as 64 bytes, so it is doubled; the calibration kernel of
tools/pmc_decode.py (an elementwise pass over 1 GiB) is listed so the correction can be checked.
"""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            key = ("paged_decode_kernel" if "paged_decode_kernel" in name else "decode_merge_kernel" if "decode_merge_kernel" in name
                   else "decode_plan_kernel" if "decode_plan_kernel" in name else "calibration_elementwise" if ("MulFunctor" in name or "mul" in name.lower()) and "vectorized" in name
                   else None)
            if key is None:
                continue
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    info = json.load(open(sys.argv[1]))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_decode.py",
           "unit": "bytes per launch", "launches": info["launches"], "avg_algorithmic_bytes": info["avg_algorithmic_bytes"]}
    fetch = per_kernel(sys.argv[2], "FETCH_SIZE")
    for k, (n, kib) in fetch.items():
        out.setdefault(k + "_detail", {})["fetch_raw_KiB_per_launch"] = kib / n
        out[k + "_detail"]["fetch_bytes_per_launch_x2_gfx950"] = kib / n * 1024 * 2
        out[k + "_detail"]["launches_seen"] = n
    if len(sys.argv) > 3:
        wr = per_kernel(sys.argv[3], "WRITE_SIZE")
        for k, (n, kib) in wr.items():
            out.setdefault(k + "_detail", {})["write_bytes_per_launch"] = kib / n * 1024
    for k in ("paged_decode_kernel", "decode_merge_kernel"):
        d = out.get(k + "_detail")
        if d:
            out[k] = int(d.get("fetch_bytes_per_launch_x2_gfx950", 0) + d.get("write_bytes_per_launch", 0))
    cal = out.get("calibration_elementwise_detail")
    if cal:
        cal["expected_bytes_each_way"] = info["calibration_bytes_each_way"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
