"""Per-launch listing of one training step: layer tag, kernel, ms, TFLOP/s, GB/s, in launch order
(HIP events around every launch on its own stream, see miseg_amd/profile.py)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd  # noqa: E402,F401
from miseg_amd import netspec, trainer, weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet18")
ap.add_argument("--decoder", default="U-Net")
ap.add_argument("--height", type=int, default=1920)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--loss", default="cce_dice")
ap.add_argument("--min-ms", type=float, default=0.0)
a = ap.parse_args()

g = netspec.build_graph(a.encoder, a.decoder, (a.height, a.width, 3), 3, "softmax", None)
tr = trainer.Trainer(g, W.init_weights(g, seed=0), a.batch, loss=a.loss)
x = torch.rand((a.batch, a.height, a.width, 3), device="cuda")
gt = torch.nn.functional.one_hot(torch.randint(0, 3, (a.batch, a.height, a.width), device="cuda"), 3).float()
tr.train_step(x, gt)
tr.profile_step(x, gt)
total = 0.0
for tag, kernel, ms, flops, nbytes in tr.lib.calls():
    total += ms
    if ms >= a.min_ms:
        print("%-34s %-46s %8.3f ms %8.1f TFLOP/s %8.1f GB/s" % (
            tag, kernel[:46], ms, flops / ms * 1e-9 if ms else 0, nbytes / ms * 1e-6 if ms else 0))
print("sum of launches: %.2f ms" % total)
