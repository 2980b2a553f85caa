# One GPU call that produces everything profiles/<tag>_* is reduced from (scripts/collect_profiles.py): the default
# bench line, per-kernel durations (headline with one and two plans in flight, DenseNet-169 LinkNet and FPN inference,
# ResNet-18 U-Net and DenseNet-169 FPN training steps), and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: each
# needs a run of its own) of the headline, DenseNet-169 and training configurations.
# usage: bash scripts/profile_round.sh [dir under gpurun_out, default r03] [traces|counters|all]
# (two calls, one per half, keep each within one GPU call's time limit)
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03}
mkdir -p $O
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
T=$GRAFT_REPO_ROOT/scripts/train_bench.py
COMMON="--no-cpu-baseline --no-train --no-extra"
ks() {  # name, then the program and its arguments; the program's stdout is kept beside the summary
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 "$@" > $O/${name}_bench.txt 2>/dev/null
  echo kernel trace $name done
}
if [ $PART != counters ]; then
cd $GRAFT_REPO_ROOT
timeout -k 10 560 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json; echo
cd /tmp
ks ks1 $B --streams 1 --steps 10 --warmup 2 $COMMON
ks ks2 $B --steps 10 --warmup 2 $COMMON
ks ks_d169_linknet $B --encoder densenet169 --decoder Linknet --batch 8 --steps 6 --warmup 2 $COMMON
ks ks_d169_fpn $B --encoder densenet169 --decoder FPN --batch 4 --steps 6 --warmup 2 $COMMON
ks kst $T --batch 16 --steps 3 --warmup 1 --loss cce_dice
ks kst_d169_fpn $T --encoder densenet169 --decoder FPN --batch 4 --steps 3 --warmup 1 --loss dice_loss
fi
[ $PART = traces ] && exit 0
pmc() {  # name, then the bench arguments
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$name -o fetch -- python3 "$@" > /dev/null 2>&1
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$name -o write -- python3 "$@" > /dev/null 2>&1
  echo pmc $name done
}
pmc r34_unet $B --steps 2 --warmup 1 $COMMON
pmc d169_linknet $B --encoder densenet169 --decoder Linknet --batch 8 --steps 2 --warmup 1 $COMMON
pmc d169_fpn $B --encoder densenet169 --decoder FPN --batch 4 --steps 2 --warmup 1 $COMMON
pmc train_r18_unet $T --batch 16 --steps 1 --warmup 1 --loss cce_dice
find $O -name "*.csv" | head -60
