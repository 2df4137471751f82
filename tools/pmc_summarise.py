#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files for the decode-attention launches.

usage: pmc_summarise.py info.json fetch.csv|dir [write.csv|dir] > profiles/pmc_traffic.json

The summary records the SHA-256 of the kernel source it was taken from (csrc/decode_attn.hip as it is in the
tree when this script runs: run it right after the counter passes) and the commit at hand; bench.py compares
the hash with the source it runs from and marks the traffic figure stale when they differ.

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte
request of a wide (16 B per lane) read as 64 bytes, so it is doubled; the calibration kernel of
tools/pmc_decode.py (an elementwise pass over 1 GiB) is listed so the correction can be checked.
"""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tables(path):
    """A counter_collection.csv, or every one below a directory."""
    if os.path.isdir(path):
        return sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
    return [path]


def source_stamp(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", rel], capture_output=True, text=True).stdout.strip())
    except OSError:
        head, dirty = "", False
    return {"file": rel, "sha256": digest, "commit": head + ("+uncommitted changes to the file" if dirty else "")}


def per_kernel(path, counter):
    acc = {}
    rows = []
    for table in tables(path):
        with open(table) as f:
            rows.extend(csv.DictReader(f))
    for r in rows:
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
           "unit": "bytes per launch", "heads (query, kv)": info.get("heads", [16, 8]), "launches": info["launches"], "avg_algorithmic_bytes": info["avg_algorithmic_bytes"],
           "kernel_source": source_stamp("nano_vllm_x_amd/csrc/decode_attn.hip")}
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
