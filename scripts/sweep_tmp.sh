set -e
for d in "4 60 80 512 512 0" "4 120 160 512 256 0" "4 240 320 256 128 0" "4 480 640 128 64 0" "4 960 1280 64 32 0"; do
  timeout -k 5 60 python scripts/conv_bench.py $d --transposed --relu --iters 30
  timeout -k 5 60 python scripts/conv_bench.py $d --transposed --relu --iters 30 --hint 64
done
