"""Training throughput of the HIP path on synthetic data (one rank)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd  # noqa: E402,F401
from miseg_amd import netspec, trainer, weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet18")
ap.add_argument("--decoder", default="U-Net")
ap.add_argument("--height", type=int, default=1920)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--loss", default="dice_loss")
ap.add_argument("--main-priority", type=int, default=None,
                help="run the step on a stream of this priority (-1 = high) instead of the default stream; "
                     "the weight-gradient stream keeps the default priority")
a = ap.parse_args()

g = netspec.build_graph(a.encoder, a.decoder, (a.height, a.width, 3), 3, "softmax", None)
w = W.init_weights(g, seed=0)
tr = trainer.Trainer(g, w, a.batch, loss=a.loss)
x = torch.rand((a.batch, a.height, a.width, 3), device="cuda")
lab = torch.randint(0, 3, (a.batch, a.height, a.width), device="cuda")
gt = torch.nn.functional.one_hot(lab, 3).float()
torch.cuda.synchronize()
print("stream priority range", torch.cuda.Stream.priority_range())
main = torch.cuda.Stream(priority=a.main_priority) if a.main_priority is not None else torch.cuda.current_stream()
with torch.cuda.stream(main):
    for _ in range(a.warmup):
        tr.train_step(x, gt)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        tr.train_step(x, gt)
    torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
print("train %s %s %dx%d batch %d: %.1f ms/step, %.2f images/s, peak mem %.1f GiB" % (
    a.encoder, a.decoder, a.height, a.width, a.batch, dt * 1e3, a.batch / dt,
    torch.cuda.max_memory_allocated() / 2 ** 30))
# phase split
torch.cuda.synchronize(); t0 = time.time(); tr.forward(x); torch.cuda.synchronize(); t1 = time.time()
tr.loss_and_metrics(gt); tr.backward(); torch.cuda.synchronize(); t2 = time.time()
tr.adam_step(); torch.cuda.synchronize(); t3 = time.time()
print("forward %.1f ms, backward %.1f ms, adam %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
