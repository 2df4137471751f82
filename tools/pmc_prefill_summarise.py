#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_all.sh over tools/pmc_prefill.py (SQ counters of the prefill
attention kernel on the bench workload's prompts).

usage: pmc_prefill_summarise.py pf_info.json pass1_dir pass2_dir [fetch_dir write_dir] > profiles/rN_pmc_prefill.json

Counter units as rocprofv3 reports them on gfx950: SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over
waves; SQ_BUSY_CYCLES per shader engine; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs.
"""
import csv
import json
import sys

from pmc_summarise import source_stamp, tables


def per_launch(path):
    acc, n = {}, {}
    for table in tables(path):
        with open(table) as f:
            for r in csv.DictReader(f):
                if "paged_prefill_kernel" not in r["Kernel_Name"]:
                    continue
                c = r["Counter_Name"]
                acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
    return {c: acc[c] / n[c] for c in acc}


def main():
    info = json.load(open(sys.argv[1]))
    c = per_launch(sys.argv[2])
    c.update(per_launch(sys.argv[3]))
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 1024          # active cycles of one XCD x 1024 SIMDs on the chip
    d = {
        "GRBM_GUI_ACTIVE_per_XCD": c["GRBM_GUI_ACTIVE"] / 8,
        "MfmaUtil_percent (MFMA busy cycles / (active cycles x 1024 SIMDs))": 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
        "valu_active_percent (4 x SQ_ACTIVE_INST_VALU / SIMD cycles)": 100 * 4 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
        "lds_active_percent (4 x SQ_ACTIVE_INST_LDS / SIMD cycles)": 100 * 4 * c["SQ_ACTIVE_INST_LDS"] / simd_cycles,
        "waves_per_simd (4 x SQ_WAVE_CYCLES / SIMD cycles)": 4 * c["SQ_WAVE_CYCLES"] / simd_cycles,
        "wait_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
        "transcendental_per_mfma": c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_INSTS_MFMA"],
        "lds_instructions_per_mfma": c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"],
        "mfma_flop (SQ_INSTS_MFMA x 32768)": c["SQ_INSTS_MFMA"] * 32768,
        "algorithmic_flop": info["algorithmic_flop_per_launch"],
    }
    if len(sys.argv) > 5:   # HBM traffic: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (see pmc_summarise.py)
        c.update(per_launch(sys.argv[4]))
        c.update(per_launch(sys.argv[5]))
        hbm = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
        d["hbm_bytes (2 x FETCH_SIZE + WRITE_SIZE)"] = hbm
        d["compulsory_bytes (q, out, K and V rows once each)"] = info.get("compulsory_bytes_per_launch")
        if info.get("compulsory_bytes_per_launch"):
            d["hbm_bytes_over_compulsory"] = hbm / info["compulsory_bytes_per_launch"]
            d["flop_per_compulsory_byte (ridge of 2.5 PFLOP/s over 8 TB/s: 312)"] = info["algorithmic_flop_per_launch"] / info["compulsory_bytes_per_launch"]
    out = {"source": "rocprofv3 --pmc (two passes, tools/pmc_all.sh) -- python3 tools/pmc_prefill.py",
           "kernel": "paged_prefill_kernel<2,4> (4 waves, 64 queries, two workgroups per CU, LDS-DMA gather)",
           "kernel_source": source_stamp("nano_vllm_x_amd/csrc/prefill_attn.hip"),
           "workload": f"bench prompts: 256 sequences, {info['tokens']} tokens, one layer, 16 q heads / 8 kv heads",
           "per_launch": c, "derived": d}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
