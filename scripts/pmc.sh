#!/bin/bash
# usage: scripts/pmc.sh <tag> <counters...> -- <conv_bench args>   (run on the GPU box)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/conv_bench.py "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_$tag/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("$tag", k, "mean %.4g over %d dispatches" % (sum(v) / len(v), len(v)))
PY
