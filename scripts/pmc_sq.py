#!/usr/bin/env python3
"""Reduce one rocprofv3 SQ counter pass (--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace) to per-kernel averages
for this library's kernels.

usage: scripts/pmc_sq.py <counter_collection.csv> <out.json>

Derived per kernel (per launch): GRBM_GUI_ACTIVE is summed over the 8 XCDs, so cycles per XCD = GRBM / 8; the SQ
counters are summed over all CUs.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles per XCD)
... as rocprofv3 reports the counter per SIMD; lds_bank_conflict_frac_of_cycles = SQ_LDS_BANK_CONFLICT /
(256 CUs x cycles per XCD); wait_frac_of_wave_cycles = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys


def main():
    src, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"]
        if "miseg::" not in name and "miseg_f16::" not in name:
            continue
        name = name.replace("void ", "").replace("miseg_f16::", "").replace("miseg::", "").split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for name, cs in acc.items():
        k = {c: sum(v) / len(v) for c, v in cs.items()}
        k["launches"] = max(len(v) for v in cs.values())
        cyc = k.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            d = {"cycles_per_xcd": cyc}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
                d["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc)
            if "SQ_LDS_BANK_CONFLICT" in k:
                d["lds_bank_conflict_frac_of_cycles"] = k["SQ_LDS_BANK_CONFLICT"] / (256 * cyc)
            if "SQ_LDS_IDX_ACTIVE" in k and k["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in k:
                d["lds_bank_conflict_frac_of_lds_active"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
                d["lds_active_frac_of_cycles"] = k["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
            if k.get("SQ_WAVE_CYCLES"):
                d["wait_frac_of_wave_cycles"] = k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
            k["derived"] = d
        kernels[name] = k
    json.dump({"kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)
    for name, k in sorted(kernels.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"]):
        d = k.get("derived", {})
        print("%-68s n=%4d  mfma %.2f  lds-active %.2f  conflicts %.3f of cycles  waiting %.2f" % (
            name[:68], k["launches"], d.get("mfma_busy_frac", 0), d.get("lds_active_frac_of_cycles", 0),
            d.get("lds_bank_conflict_frac_of_cycles", 0), d.get("wait_frac_of_wave_cycles", 0)))


if __name__ == "__main__":
    main()
