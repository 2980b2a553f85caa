#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; one pass each, since both do not fit
the four TCC counters of one pass) into HBM bytes per launch for every kernel of this library.

usage: scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections: both counters are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide streaming reads at 64 bytes, so the read side is doubled before it is compared with a
byte count. WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if r["Counter_Name"] != counter or ("miseg::" not in name and "miseg_f16::" not in name):
            continue
        # the fp16-storage compilation of a kernel (namespace miseg_f16) is reported under the same name
        name = name.replace("void ", "").replace("miseg_f16::", "").replace("miseg::", "").split("(")[0]
        acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for name in f:
        fk = sum(f[name]) / len(f[name])
        wk = sum(w[name]) / len(w[name]) if name in w else 0.0
        rd = fk * 1024.0 * 2.0
        wr = wk * 1024.0
        kernels[name] = {"dispatches": len(f[name]), "fetch_size_kib_raw": fk, "write_size_kib": wk,
                         "hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd,
                         "write_bytes_per_launch": wr}
    json.dump({"units": "FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of "
                        "the bytes of wide streaming reads and is doubled here",
               "kernels": kernels}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
