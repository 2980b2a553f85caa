# Transpose-conv layer sweep (decoder shapes of the ResNet-34 U-Net at 1920x2560, batch 4): the wave-private-ring
# kernel against the block-shared-weights kernel (--hint 64).  Run on a GPU box from the repo root.
set -e
for d in "4 60 80 512 512 0" "4 120 160 512 256 0" "4 240 320 256 128 0" "4 480 640 128 64 0" "4 960 1280 64 32 0"; do
  timeout -k 5 60 python scripts/conv_bench.py $d --transposed --relu --iters 30
  timeout -k 5 60 python scripts/conv_bench.py $d --transposed --relu --iters 30 --hint 64
done
