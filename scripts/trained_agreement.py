"""Mask agreement of the inference engine with the fp32 oracle on TRAINED weights: a ResNet-34 U-Net is trained
with the HIP trainer for a few hundred steps on synthetic blocky masks (class-coloured blocks + noise), then one
fresh 1920x2560 image goes through the engine (fp16 and bf16 storage) and through oracle/nets.py on the host.
The margins of a trained network are wider than those of the seeded random-weight networks the tests use.

usage: python scripts/trained_agreement.py [--steps 300] [--batch 2] [--out profiles/<name>.txt]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd  # noqa: E402,F401
from miseg_amd import engine, netspec, trainer, weights as W, losses as hip_losses  # noqa: E402
from oracle import nets, ops_torch, post, losses  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet34")
ap.add_argument("--decoder", default="U-Net")
ap.add_argument("--height", type=int, default=1920)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--block", type=int, default=64)
ap.add_argument("--noise", type=float, default=0.2)
ap.add_argument("--out", default=None)
a = ap.parse_args()
h, w, n, classes = a.height, a.width, a.batch, 3
dev = torch.device("cuda:0")
palette = torch.tensor([[0.15, 0.2, 0.6], [0.8, 0.3, 0.25], [0.45, 0.75, 0.4]], device=dev)


def batch(gen, count):
    """Blocky class field (p = 0.7 / 0.2 / 0.1), class-coloured image + noise, per-image min-max (the reference's
    normalize_image, utils/data.py:64-80), one-hot target."""
    u = torch.rand((count, h // a.block, w // a.block), device=dev, generator=gen)
    ids = (u > 0.7).long() + (u > 0.9).long()
    ids = ids.repeat_interleave(a.block, 1).repeat_interleave(a.block, 2)
    img = palette[ids] + a.noise * torch.randn((count, h, w, 3), device=dev, generator=gen)
    lo = img.amin(dim=(1, 2, 3), keepdim=True)
    hi = img.amax(dim=(1, 2, 3), keepdim=True)
    img = (img - lo) / (hi - lo)
    return img.contiguous(), torch.nn.functional.one_hot(ids, classes).float(), ids


lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


g = netspec.build_graph(a.encoder, a.decoder, (h, w, 3), classes, "softmax", None)
tr = trainer.Trainer(g, W.init_weights(g, seed=1234), n, loss="cce_dice", learning_rate=a.lr)
gen = torch.Generator(device=dev).manual_seed(99)
say("%s %s %dx%d: %d training steps of %d images (cce+Dice, Adam lr %g) on blocky masks, noise %.2f" % (
    a.encoder, a.decoder, h, w, a.steps, n, a.lr, a.noise))
t0 = time.time()
for step in range(a.steps):
    x, gt, _ = batch(gen, n)
    sums, pixels = tr.train_step(x, gt)
    if step % 50 == 0 or step == a.steps - 1:
        s = sums.cpu().numpy().reshape(-1, classes)
        say("  step %4d: loss %.4f  f1 %.4f" % (step, hip_losses.cce_dice_loss.from_sums(s, pixels),
                                                 hip_losses.f1_score.from_sums(s, pixels)))
torch.cuda.synchronize()
say("  trained in %.1f s" % (time.time() - t0))
wts = tr.get_weights()
del tr
torch.cuda.empty_cache()

xe, gte, ids = batch(torch.Generator(device=dev).manual_seed(7), 1)
x_np = xe.cpu().numpy()
gt_np = gte.cpu().numpy()
with torch.no_grad():
    want = ops_torch.to_numpy(nets.forward(ops_torch, a.encoder, a.decoder, wts, x_np))
m_want = np.stack([post.collapse_probabilities(p.copy()) for p in want])
acc = float((want.argmax(-1) == ids.cpu().numpy()).mean())
srt = np.sort(want, axis=-1)
margin = srt[..., -1] - srt[..., -2]
say("oracle (fp32, host): pixel accuracy against the ground truth %.4f, f1 %.4f; top-2 margin below 1e-2 on %.4f %% of "
    "the pixels, below 1e-3 on %.4f %%" % (acc, losses.f1_score(gt_np, want), 100 * float((margin < 1e-2).mean()),
                                           100 * float((margin < 1e-3).mean())))
for storage in ("fp16", "bf16"):
    eng = engine.Engine(g, wts, batch=1, storage=storage)
    got = eng.forward(xe).cpu().numpy()
    m_got = eng.mask.cpu().numpy()
    del eng
    torch.cuda.empty_cache()
    agree = float((m_got == m_want).all(axis=-1).mean())
    say("%s storage: masks equal to the oracle's on %.4f %% of the pixels, max |dp| %.2e, |dF1| %.2e, |dIoU| %.2e" % (
        storage, 100 * agree, float(np.abs(got - want).max()), abs(losses.f1_score(gt_np, got) - losses.f1_score(gt_np, want)),
        abs(losses.iou_score(gt_np, got) - losses.iou_score(gt_np, want))))
if a.out:
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
