#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag>
# Writes the bench line, the rocprofv3 kernel-stats summary of the same command and the two HBM
# counter passes (FETCH_SIZE, WRITE_SIZE: one pass each) under gpurun_out/<tag>_*.
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${tag}_stats_bench.json 2> gpurun_out/${tag}_stats.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> gpurun_out/${tag}_fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2> gpurun_out/${tag}_write.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats1 -o s -- python3 bench.py --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-train > gpurun_out/${tag}_stats1_bench.json 2> gpurun_out/${tag}_stats1.log
ls gpurun_out/${tag}_*
