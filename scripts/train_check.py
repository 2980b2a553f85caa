"""Debug aid: one training step of the HIP trainer against the torch-CPU oracle, per-layer errors."""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd as pkg                      # noqa: E402
from miseg_amd import netspec, trainer       # noqa: E402
from oracle import nets, train               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="resnet18")
ap.add_argument("--decoder", default="U-Net")
ap.add_argument("--h", type=int, default=64)
ap.add_argument("--w", type=int, default=96)
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--classes", type=int, default=3)
ap.add_argument("--fp32", action="store_true", help="compare with the fp32 oracle instead of the bf16-storage one")
a = ap.parse_args()

spec = nets.build_spec(a.encoder, a.decoder, a.classes)
w = nets.init_weights(spec, 0, True)
rng = np.random.default_rng(1)
x = rng.random((a.n, a.h, a.w, 3), dtype=np.float32)
gt = np.eye(a.classes, dtype=np.float32)[rng.integers(0, a.classes, (a.n, a.h, a.w))]
from oracle import ops_torch, ops_bf16
OPS = ops_torch if a.fp32 else ops_bf16
loss, pr, grads, bs = train.loss_and_grads(a.encoder, a.decoder, w, x, gt, ops=OPS)
print("oracle loss", loss)
taps = {}
nets.forward(OPS, a.encoder, a.decoder, w, x, training=True, taps=taps)

g = netspec.build_graph(a.encoder, a.decoder, (a.h, a.w, 3), a.classes, "softmax", None)
tr = trainer.Trainer(g, w, a.n)
xd = torch.from_numpy(x).cuda()
gd = torch.from_numpy(gt).cuda()
probs = tr.forward(xd)
torch.cuda.synchronize()
print("probs err", np.abs(probs.cpu().numpy() - pr).max())
for name, ref in taps.items():
    if name in tr.acts:
        mine = tr.acts[name].float().cpu().numpy()
        print("tap %-30s relL2 %.4f" % (name, np.linalg.norm(mine - ref) / (np.linalg.norm(ref) + 1e-12)))
sums, pixels = tr.loss_and_metrics(gd)
tr.keep_grads = True
tr.backward()
torch.cuda.synchronize()
# batch stats
for name, (m, v) in bs.items():
    s = tr.bn_state[name]
    c = m.shape[0]
    em = np.abs(s["mean"][:c].cpu().numpy() - m).max() / (np.abs(m).max() + 1e-6)
    ev = np.abs(s["var"][:c].cpu().numpy() - v).max() / (np.abs(v).max() + 1e-6)
    if em > 0.05 or ev > 0.05:
        print("BN stat", name, em, ev)
got = {}
P = tr.P
tr.P = tr.G                                   # read the gradient buffer through the weight views
gw = tr.get_weights()
tr.P = P
worst = 0
for lname, d in grads.items():
    for n, ref in d.items():
        if n not in gw[lname]:
            continue
        mine = gw[lname][n]
        den = np.linalg.norm(ref) + 1e-12
        err = np.linalg.norm(mine - ref) / den
        worst = max(worst, err)
        flag = "" if err < 0.05 else "   <<<<"
        print("%-34s %-7s |ref| %.3e relL2 %.4f%s" % (lname, n, den, err, flag))
print("worst", worst)

# ---- per-op forward self-check: each conv / BN recomputed with torch from the trainer's own inputs
import torch.nn.functional as F
kw = {k: v for k, v in w.items()}
for name, l in g.layers.items():
    if l.cls == "Conv2D" and name in tr.acts and l.config["kernel_size"] != 7:
        xin, pad = tr.saved[name]
        xf = xin.float().cpu().permute(0, 3, 1, 2)
        wt = torch.from_numpy(kw[name]["kernel"]).to(torch.bfloat16).float().permute(3, 2, 0, 1)
        b = torch.from_numpy(kw[name]["bias"]) if "bias" in kw[name] else None
        ref = F.conv2d(xf, wt, b, stride=l.config["strides"], padding=pad).permute(0, 2, 3, 1)
        ref = ref.to(torch.bfloat16).float()
        mine = tr.acts[name].float().cpu()
        d = (mine - ref).abs()
        idx = np.unravel_index(int(d.argmax()), d.shape)
        print("op %-28s relL2 %.5f  maxabs %.4f at %s (ref %.4f)" % (
            name, float((mine - ref).norm() / ref.norm()), float(d.max()), idx, float(ref[idx])))
    if l.cls == "BatchNormalization" and name in tr.saved:
        xin = tr.saved[name]
        xf = xin.float().cpu()
        mean = xf.mean(dim=(0, 1, 2))
        var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
        gam = torch.from_numpy(kw[name]["gamma"]) if "gamma" in kw[name] else 1.0
        ref = torch.relu((xf - mean) * torch.rsqrt(var + l.config["epsilon"]) * gam + torch.from_numpy(kw[name]["beta"]))
        ref = ref.to(torch.bfloat16).float()
        mine = tr.acts[tr.consumers[name][0]].float().cpu()
        d = (mine - ref).abs()
        print("op %-28s relL2 %.5f  maxabs %.4f" % (name, float((mine - ref).norm() / ref.norm()), float(d.max())))
for name, l in g.layers.items():
    if l.cls == "Add":
        a_, b_ = tr.acts[l.inbound[0]].float().cpu(), tr.acts[l.inbound[1]].float().cpu()
        ref = (a_ + b_).to(torch.bfloat16).float()
        mine = tr.acts[name].float().cpu()
        print("op %-28s relL2 %.5f  maxabs %.4f" % (name, float((mine - ref).norm() / ref.norm()), float((mine - ref).abs().max())))
    if l.cls == "Conv2DTranspose":
        xin, _ = tr.saved[name]
        xf = xin.float().cpu().permute(0, 3, 1, 2)
        wt = torch.from_numpy(kw[name]["kernel"]).to(torch.bfloat16).float().permute(3, 2, 0, 1)
        ref = F.conv_transpose2d(xf, wt, stride=2, padding=1).permute(0, 2, 3, 1).to(torch.bfloat16).float()
        mine = tr.acts[name].float().cpu()
        print("op %-28s relL2 %.5f  maxabs %.4f" % (name, float((mine - ref).norm() / ref.norm()), float((mine - ref).abs().max())))
    if l.cls == "MaxPooling2D":
        xin = tr.saved[name]
        ref = F.max_pool2d(F.pad(xin.float().cpu().permute(0, 3, 1, 2), (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
        mine = tr.acts[name].float().cpu()
        print("op %-28s relL2 %.5f  maxabs %.4f" % (name, float((mine - ref).norm() / ref.norm()), float((mine - ref).abs().max())))
    if l.cls == "Concatenate":
        ref = torch.cat([tr.acts[i].float().cpu() for i in l.inbound], dim=-1)
        mine = tr.acts[name].float().cpu()
        print("op %-28s relL2 %.5f  maxabs %.4f" % (name, float((mine - ref).norm() / ref.norm()), float((mine - ref).abs().max())))

# ---- per-op backward self-check against torch autograd on the trainer's own inputs
def rel(a_, b_):
    return float((a_ - b_).norm() / (b_.norm() + 1e-20))

for name, l in g.layers.items():
    if name not in tr.grad_log and not (l.cls == "Conv2D" and name in tr.saved):
        continue
    if l.cls == "Conv2D" and l.config["kernel_size"] != 7 and name in tr.grad_log:
        dy, dx = tr.grad_log[name]
        xin, pad = tr.saved[name]
        xf = xin.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
        wt = torch.from_numpy(kw[name]["kernel"]).to(torch.bfloat16).float().permute(3, 2, 0, 1).clone().requires_grad_(True)
        y = F.conv2d(xf, wt, None, stride=l.config["strides"], padding=pad)
        y.backward(dy.float().cpu().permute(0, 3, 1, 2))
        print("bw %-28s dx %.5f  dw %.5f" % (name, rel(dx.float().cpu(), xf.grad.permute(0, 2, 3, 1)),
              rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
    elif l.cls == "Conv2DTranspose":
        dy, dx = tr.grad_log[name]
        xin, _ = tr.saved[name]
        xf = xin.float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
        wt = torch.from_numpy(kw[name]["kernel"]).to(torch.bfloat16).float().permute(3, 2, 0, 1).clone().requires_grad_(True)
        y = F.conv_transpose2d(xf, wt, stride=2, padding=1)
        y.backward(dy.float().cpu().permute(0, 3, 1, 2))
        print("bw %-28s dx %.5f  dw %.5f" % (name, rel(dx.float().cpu(), xf.grad.permute(0, 2, 3, 1)),
              rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
    elif l.cls == "BatchNormalization" and name in tr.grad_log:
        dy, dx = tr.grad_log[name]
        xin = tr.saved[name]
        xf = xin.float().cpu().clone().requires_grad_(True)
        gam = torch.from_numpy(kw[name]["gamma"]).clone().requires_grad_(True)
        bet = torch.from_numpy(kw[name]["beta"]).clone().requires_grad_(True)
        mean = xf.mean(dim=(0, 1, 2))
        var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
        y = torch.relu((xf - mean) * torch.rsqrt(var + l.config["epsilon"]) * gam + bet)
        y.backward(dy.float().cpu())
        print("bw %-28s dx %.5f  dgamma %.5f dbeta %.5f" % (name, rel(dx.float().cpu(), xf.grad),
              rel(torch.from_numpy(gw[name]["gamma"]), gam.grad), rel(torch.from_numpy(gw[name]["beta"]), bet.grad)))
    elif l.cls == "MaxPooling2D":
        dy, dx = tr.grad_log[name]
        xf = tr.saved[name].float().cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
        y = F.max_pool2d(F.pad(xf, (1, 1, 1, 1)), 3, 2)
        y.backward(dy.float().cpu().permute(0, 3, 1, 2))
        print("bw %-28s dx %.5f" % (name, rel(dx.float().cpu(), xf.grad.permute(0, 2, 3, 1))))

# ---- stem (input BatchNorm + 7x7 stride-2 convolution) backward
stem = [n for n, l in g.layers.items() if l.cls == "Conv2D" and l.config["kernel_size"] == 7][0]
dy, _ = tr.grad_log[stem]
xf = torch.from_numpy(x)
wt = torch.from_numpy(kw[stem]["kernel"]).to(torch.bfloat16).float().permute(3, 2, 0, 1).clone().requires_grad_(True)
if "bn_data" in kw:
    bet = torch.from_numpy(kw["bn_data"]["beta"]).clone().requires_grad_(True)
    mean = xf.mean(dim=(0, 1, 2))
    var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
    xn = (xf - mean) * torch.rsqrt(var + 2e-5) + bet
else:
    xn = xf
xn = ops_bf16.r(xn)
y = F.conv2d(xn.permute(0, 3, 1, 2), wt, None, stride=2, padding=3)
y.backward(dy.float().cpu().permute(0, 3, 1, 2))
print("bw %-28s dw %.5f" % (stem, rel(torch.from_numpy(gw[stem]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
if "bn_data" in kw:
    print("bw bn_data dbeta mine", gw["bn_data"]["beta"], "autograd", bet.grad.numpy())
