#!/usr/bin/env python3
"""Copy what scripts/profile_round.sh left under gpurun_out/r03 into profiles/ under this round's names:
the bench lines, the per-kernel duration summaries as rocprofv3 wrote them, the two counter passes of every
configuration trimmed to this library's kernels and the columns that matter, and their reduction to HBM bytes
per launch (scripts/pmc_traffic.py).

usage: scripts/collect_profiles.py [gpurun_out/r03] [r03]"""
import csv
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03")
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
dst = os.path.join(ROOT, "profiles")
COLS = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name",
        "Counter_Value"]


def copy(a, b):
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
    print("profiles/%s_%s" % (tag, b))


def trim(a, b):
    out = os.path.join(dst, "%s_%s" % (tag, b))
    with open(os.path.join(src, a)) as f, open(out, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(COLS)
        for r in csv.DictReader(f):
            if "miseg" in r["Kernel_Name"]:
                w.writerow([r[c] for c in COLS])
    print("profiles/%s_%s" % (tag, b))
    return out


copy("bench.json", "bench.json")
copy("ks1/ks1_kernel_stats.csv", "kernel_stats_streams1.csv")
copy("ks1_bench.txt", "kernel_stats_streams1_bench.json")
copy("ks2/ks2_kernel_stats.csv", "kernel_stats_streams2.csv")
copy("ks2_bench.txt", "kernel_stats_streams2_bench.json")
for name in ("d169_linknet", "d169_fpn"):
    copy("ks_%s/ks_%s_kernel_stats.csv" % (name, name), "kernel_stats_%s.csv" % name)
    copy("ks_%s_bench.txt" % name, "kernel_stats_%s_bench.json" % name)
copy("kst/kst_kernel_stats.csv", "train_kernel_stats_batch16.csv")
copy("kst_bench.txt", "train_kernel_stats_batch16_bench.txt")
copy("kst_d169_fpn/kst_d169_fpn_kernel_stats.csv", "train_kernel_stats_d169_fpn_batch4.csv")
copy("kst_d169_fpn_bench.txt", "train_kernel_stats_d169_fpn_batch4_bench.txt")
for cfg, suffix in (("r34_unet", ""), ("d169_linknet", "_d169_linknet"), ("d169_fpn", "_d169_fpn"),
                    ("train_r18_unet", "_train_r18_unet")):
    f = trim("fetch_%s/fetch_counter_collection.csv" % cfg, "pmc_fetch_size%s.csv" % suffix)
    w = trim("write_%s/write_counter_collection.csv" % cfg, "pmc_write_size%s.csv" % suffix)
    out = os.path.join(dst, "%s_pmc_traffic%s.json" % (tag, suffix))
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py"), f, w, out], check=True)
    print("profiles/%s_pmc_traffic%s.json" % (tag, suffix))
