set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 400 python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
tail -c 600 gpurun_out/r02b/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b/ks1 -o ks1 -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r02b/ks1_bench.json 2>/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b/ks2 -o ks2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-extra > $GRAFT_REPO_ROOT/gpurun_out/r02b/ks2_bench.json 2>/dev/null
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b/fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extra > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b/write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-extra > /dev/null 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/r02b | head -40
