# One GPU call that produces everything profiles/r02_d_* is reduced from: the default bench line, per-kernel
# durations (one and two plans in flight, training step), and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: each
# needs a run of its own) of the headline, DenseNet-169 and training configurations.
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--no-cpu-baseline --no-train --no-extra"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -o ks1 -- python3 $B --streams 1 --steps 10 --warmup 2 $COMMON > $O/ks1_bench.json 2>/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks2 -o ks2 -- python3 $B --steps 10 --warmup 2 $COMMON > $O/ks2_bench.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kst -o kst -- python3 $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 16 --steps 3 --warmup 1 --loss cce_dice > $O/kst_bench.txt 2>/dev/null
echo kernel traces done
pmc() {  # name, then the bench arguments
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$name -o fetch -- python3 "$@" > /dev/null 2>&1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$name -o write -- python3 "$@" > /dev/null 2>&1
  echo pmc $name done
}
pmc r34_unet $B --steps 2 --warmup 1 $COMMON
pmc d169_linknet $B --encoder densenet169 --decoder Linknet --batch 8 --steps 2 --warmup 1 $COMMON
pmc d169_fpn $B --encoder densenet169 --decoder FPN --batch 4 --steps 2 --warmup 1 $COMMON
pmc train_r18_unet $GRAFT_REPO_ROOT/scripts/train_bench.py --batch 16 --steps 1 --warmup 1 --loss cce_dice
find $O -name "*.csv" | head -40
