"""Layer-by-layer check of the HIP training step: every forward and backward op of a network is
recomputed with torch (CPU, fp32 math, autograd) from the very tensors the trainer fed to its own
kernel, so the comparison is free of the error amplification that a deep batch-normalised
network adds to any end-to-end comparison of two bf16-storage runs.  Used by the GPU tests and by
scripts/train_check.py."""
import numpy as np
import torch
import torch.nn.functional as F


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


def _bf(x):
    return x.to(torch.bfloat16).float()


def grads_as_weights(tr):
    """The flat gradient buffer read through the Keras-layout weight views."""
    P = tr.P
    tr.P = tr.G
    try:
        return tr.get_weights()
    finally:
        tr.P = P


def forward_errors(tr, weights):
    g = tr.g
    out = []

    def conv_ref(name, plain=False):
        l = g.layers[name]
        fs = getattr(tr, "fs", None)
        if fs is not None and name == fs["conv"]:
            # the split final_stage.  Its operands are rounded where the split puts them: the coarse part's
            # phase weights are sums of 3x3 taps rounded to bf16 AFTER the sum, and the coarse part is stored
            # (bf16) before it is added.  `plain` = the layer as the graph states it.
            k = torch.from_numpy(weights[name]["kernel"])                       # (3, 3, cin, co) fp32
            xs = [tr.acts[n_].float().cpu().permute(0, 3, 1, 2) for n_ in [fs["fine"]] + fs["srcs"]]
            c0 = xs[0].shape[1]
            if plain:
                ups = [xs[0]] + [F.interpolate(x, scale_factor=2 << i, mode="nearest") for i, x in enumerate(xs[1:])]
                return F.conv2d(torch.cat(ups, dim=1), _bf(k).permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1)
            half = torch.cat([xs[1]] + [F.interpolate(x, scale_factor=2 << i, mode="nearest")
                                        for i, x in enumerate(xs[2:])], dim=1)
            hp = F.pad(half, (1, 1, 1, 1))
            h, w = half.shape[2:]
            coarse = torch.zeros((half.shape[0], k.shape[3], 2 * h, 2 * w))
            rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
            for a in range(2):
                for b in range(2):
                    w2 = torch.zeros((k.shape[3], k.shape[2] - c0, 2, 2))
                    for dy in range(2):
                        for dx in range(2):
                            w2[:, :, dy, dx] = _bf(sum(k[ky, kx, c0:, :] for ky in rows[a][dy]
                                                       for kx in rows[b][dx])).t()
                    coarse[:, :, a::2, b::2] = F.conv2d(hp, w2)[:, :, a:a + h, b:b + w]
            fine = F.conv2d(xs[0], _bf(k[:, :, :c0, :]).permute(3, 2, 0, 1), None, padding=1)
            return (fine + _bf(coarse)).permute(0, 2, 3, 1)
        p2 = getattr(tr, "p2", None)
        if p2 is not None and name == p2["conv"]:
            # the folded finest FPN level: 3x3 convolution of the skip tensor with the composed kernel plus the
            # phase convolution of q3 = p3 + bias; operands rounded where the fold rounds them.  `plain` = the
            # layers as the graph states them (lateral 1x1 + bias, up-sampling, Add, 3x3), each stored in bf16.
            k3 = _bf(torch.from_numpy(weights[name]["kernel"]))                   # (3, 3, cp, co)
            kl = _bf(torch.from_numpy(weights[p2["lat"]]["kernel"]))[0, 0]        # (cs, cp)
            bl = torch.from_numpy(weights[p2["lat"]]["bias"])
            p3 = tr.acts[p2["coarse"]].float().cpu().permute(0, 3, 1, 2)
            sk = tr.acts[p2["skip"]].float().cpu().permute(0, 3, 1, 2)
            if plain:
                lat = _bf(F.conv2d(sk, kl.t()[:, :, None, None], bl))
                x = _bf(F.interpolate(p3, scale_factor=2, mode="nearest") + lat)
                return F.conv2d(x, k3.permute(3, 2, 0, 1), None, padding=1).permute(0, 2, 3, 1)
            q3 = _bf(p3 + bl[None, :, None, None])
            hp = F.pad(q3, (1, 1, 1, 1))
            h, w = q3.shape[2:]
            coarse = torch.zeros((q3.shape[0], k3.shape[3], 2 * h, 2 * w))
            rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
            for a in range(2):
                for b in range(2):
                    w2 = torch.zeros((k3.shape[3], k3.shape[2], 2, 2))
                    for dy in range(2):
                        for dx in range(2):
                            w2[:, :, dy, dx] = _bf(sum(k3[ky, kx] for ky in rows[a][dy] for kx in rows[b][dx])).t()
                    coarse[:, :, a::2, b::2] = F.conv2d(hp, w2)[:, :, a:a + h, b:b + w]
            wc = _bf(torch.einsum("sc,yxco->yxso", kl, k3))                       # (3, 3, cs, co)
            fine = F.conv2d(sk, wc.permute(3, 2, 0, 1), None, padding=1)
            return (fine + _bf(coarse)).permute(0, 2, 3, 1)
        xin, pad = tr.saved[name]
        wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1)
        b = torch.from_numpy(weights[name]["bias"]) if "bias" in weights[name] else None
        ref = F.conv2d(xin.float().cpu().permute(0, 3, 1, 2), wt, b, stride=l.config["strides"],
                       padding=pad).permute(0, 2, 3, 1)
        return torch.relu(ref) if l.config.get("activation") == "relu" else ref
    for name, l in g.layers.items():
        if l.cls == "Conv2D" and name in tr.acts and l.config["kernel_size"] != 7:
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), _bf(conv_ref(name)))))
            if (getattr(tr, "fs", None) and name == tr.fs["conv"]) or (getattr(tr, "p2", None) and name == tr.p2["conv"]):
                # against the layer as stated the split differs by the rounding of its weights (2^-9 per
                # phase weight, as for any bf16 weight): a wider bar applies to this entry
                out.append((name, "fwd-plain", _rel(tr.acts[name].float().cpu(), _bf(conv_ref(name, True)))))
        elif l.cls == "Conv2D" and name == g.logits_layer and g.activation == "softmax":
            # the head's probabilities; when the trainer runs final_upsampling + head_conv in their
            # half-resolution form, the reference is still the plain order of operations
            xin, pad = tr.saved[name]
            xf = xin.float().cpu().permute(0, 3, 1, 2)
            if getattr(tr, "head_up", None):
                xf = F.interpolate(xf, scale_factor=2, mode="bilinear", align_corners=False)   # never stored
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1)
            logits = F.conv2d(xf, wt, torch.from_numpy(weights[name]["bias"]), padding=pad).permute(0, 2, 3, 1)
            ref = torch.softmax(logits / (g.temperature or 1.0), dim=-1)
            out.append((name, "fwd", _rel(tr.probs.float().cpu(), ref)))
        elif l.cls == "Conv2DTranspose":
            xin, _ = tr.saved[name]
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1)
            ref = F.conv_transpose2d(xin.float().cpu().permute(0, 3, 1, 2), wt, stride=2, padding=1)
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), _bf(ref.permute(0, 2, 3, 1)))))
        elif l.cls == "BatchNormalization" and name in tr.saved:
            xf = tr.saved[name].float().cpu()
            mean = xf.mean(dim=(0, 1, 2))
            var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
            gam = torch.from_numpy(weights[name]["gamma"]) if "gamma" in weights[name] else 1.0
            ref = torch.relu((xf - mean) * torch.rsqrt(var + l.config["epsilon"]) * gam
                             + torch.from_numpy(weights[name]["beta"]))
            out.append((name, "fwd", _rel(tr.acts[tr.consumers[name][0]].float().cpu(), _bf(ref))))
        elif l.cls == "Add" and name in tr.acts:
            # an input the trainer never stored is a convolution whose epilogue did the addition: its
            # (unrounded) result enters the reference instead
            def part(i):
                if i in tr.acts:
                    return tr.acts[i].float().cpu()
                li = g.layers[i]
                if li.cls == "UpSampling2D":      # fused as well: the kernel adds while it up-samples
                    xf = tr.acts[li.inbound[0]].float().cpu().permute(0, 3, 1, 2)
                    return F.interpolate(xf, scale_factor=li.config["size"], mode="nearest").permute(0, 2, 3, 1)
                return conv_ref(i)
            parts = [part(i) for i in l.inbound]
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), _bf(parts[0] + parts[1]))))
        elif l.cls == "Concatenate" and name in tr.acts:
            ref = torch.cat([tr.acts[i].float().cpu() for i in l.inbound], dim=-1)
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), ref)))
        elif l.cls == "MaxPooling2D":
            xf = tr.saved[name].float().cpu().permute(0, 3, 1, 2)
            if l.config["pool_size"] == 2:
                ref = F.max_pool2d(xf, 2, 2).permute(0, 2, 3, 1)
            else:
                ref = F.max_pool2d(F.pad(xf, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), ref)))
        elif l.cls == "AveragePooling2D":
            xf = tr.acts[l.inbound[0]].float().cpu().permute(0, 3, 1, 2)
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), _bf(F.avg_pool2d(xf, 2, 2).permute(0, 2, 3, 1)))))
        elif l.cls == "UpSampling2D" and name in tr.acts:
            xf = tr.acts[l.inbound[0]].float().cpu().permute(0, 3, 1, 2)
            if l.config["interpolation"] == "bilinear":
                ref = F.interpolate(xf, scale_factor=2, mode="bilinear", align_corners=False)
            else:
                ref = F.interpolate(xf, scale_factor=l.config["size"], mode="nearest")
            out.append((name, "fwd", _rel(tr.acts[name].float().cpu(), _bf(ref.permute(0, 2, 3, 1)))))
    return out


def _masked(tr, name, xin, grad):
    """A data gradient whose kernel also applied the mask of the ReLU in front of the layer (the layer reads
    relu(bn(x)), Trainer.bwd_masked): the reference gradient with that mask on its leading channels."""
    mc = getattr(tr, "bwd_masked", {}).get(name)
    if not mc:
        return grad
    grad = grad.clone()
    grad[..., :mc] *= (xin.float().cpu()[..., :mc] > 0).float()
    return grad


def backward_errors(tr, weights, x_input):
    """Requires tr.keep_grads = True during backward()."""
    g = tr.g
    gw = grads_as_weights(tr)
    out = []
    nchw = lambda t: t.permute(0, 3, 1, 2)      # noqa: E731
    nhwc = lambda t: t.permute(0, 2, 3, 1)      # noqa: E731
    for name, l in g.layers.items():
        if name not in tr.grad_log:
            continue
        dy, dx = tr.grad_log[name]
        dyf = dy.float().cpu()
        fs = getattr(tr, "fs", None)
        if fs is not None and name == fs["conv"]:
            # the split final_stage against the plain layer: gradients at s2 (dx), at the three coarser
            # tensors (through their nearest up-sampling) and of the whole kernel
            xs = [nchw(tr.acts[n_].float().cpu()).clone().requires_grad_(True) for n_ in [fs["fine"]] + fs["srcs"]]
            ups = [xs[0]] + [F.interpolate(x, scale_factor=2 << i, mode="nearest") for i, x in enumerate(xs[1:])]
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            F.conv2d(torch.cat(ups, dim=1), wt, None, padding=1).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(xs[0].grad))))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
            for n_, x in zip(fs["srcs"], xs[1:]):
                parts = tr.accum_log[n_]
                assert len(parts) == 1
                out.append((name + "->" + n_, "dx", _rel(parts[0].float().cpu(), nhwc(x.grad))))
        elif getattr(tr, "p2", None) and name == tr.p2["conv"]:
            # the folded finest FPN level against the layers as the graph states them
            p2 = tr.p2
            p3 = nchw(tr.acts[p2["coarse"]].float().cpu()).clone().requires_grad_(True)
            sk = nchw(tr.acts[p2["skip"]].float().cpu()).clone().requires_grad_(True)
            w3 = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            wl = _bf(torch.from_numpy(weights[p2["lat"]]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            bl = torch.from_numpy(weights[p2["lat"]]["bias"]).clone().requires_grad_(True)
            # q3 = p3 + bias is stored (bf16) by the fold, as p2 is by the plain layers: the same rounding
            # here, with a straight-through gradient
            q3 = p3 + bl[None, :, None, None]
            q3 = q3 + (_bf(q3) - q3).detach()
            x = F.interpolate(q3, scale_factor=2, mode="nearest") + F.conv2d(sk, wl, None)
            F.conv2d(x, w3, None, padding=1).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(sk.grad))))
            out.append((name + "->" + p2["coarse"], "dx", _rel(tr.grad_log[p2["add"]][1].float().cpu(), nhwc(p3.grad))))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), w3.grad.permute(2, 3, 1, 0))))
            out.append((p2["lat"], "dw", _rel(torch.from_numpy(gw[p2["lat"]]["kernel"]), wl.grad.permute(2, 3, 1, 0))))
            out.append((p2["lat"], "dbias", _rel(torch.from_numpy(gw[p2["lat"]]["bias"]), bl.grad)))
        elif l.cls == "Conv2D" and name == g.logits_layer and getattr(tr, "head_up", None):
            # final_upsampling + head_conv run in their half-resolution form: dx is the gradient at the
            # up-sampling's input
            xin, pad = tr.saved[name]
            xf = nchw(xin.float().cpu()).clone().requires_grad_(True)
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            b = torch.from_numpy(weights[name]["bias"]).clone().requires_grad_(True)
            up = F.interpolate(xf, scale_factor=2, mode="bilinear", align_corners=False)
            F.conv2d(up, wt, b, padding=pad).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(xf.grad))))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
            out.append((name, "dbias", _rel(torch.from_numpy(gw[name]["bias"]), b.grad)))
        elif l.cls == "Conv2D" and l.config["kernel_size"] != 7:
            xin, pad = tr.saved[name]
            xf = nchw(xin.float().cpu()).clone().requires_grad_(True)
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            b = (torch.from_numpy(weights[name]["bias"]).clone().requires_grad_(True)
                 if "bias" in weights[name] else None)
            F.conv2d(xf, wt, b, stride=l.config["strides"], padding=pad).backward(nchw(dyf))
            if dx is not None:
                out.append((name, "dx", _rel(dx.float().cpu(), _masked(tr, name, xin, nhwc(xf.grad)))))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
            if b is not None:
                out.append((name, "dbias", _rel(torch.from_numpy(gw[name]["bias"]), b.grad)))
        elif l.cls == "Conv2D":
            # stem: input BatchNorm (when present) + 7x7 stride-2 convolution
            xf = torch.from_numpy(x_input)
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            bet = None
            if "bn_data" in weights:
                bet = torch.from_numpy(weights["bn_data"]["beta"]).clone().requires_grad_(True)
                mean = xf.mean(dim=(0, 1, 2))
                var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
                xf = (xf - mean) * torch.rsqrt(var + g.layers["bn_data"].config["epsilon"]) + bet
            xr = xf + (_bf(xf) - xf).detach()           # bf16 storage, straight-through gradient
            F.conv2d(nchw(xr), wt, None, stride=2, padding=3).backward(nchw(dyf))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
            if bet is not None:
                out.append(("bn_data", "dbeta", _rel(torch.from_numpy(gw["bn_data"]["beta"]), bet.grad)))
        elif l.cls == "Conv2DTranspose":
            xin, _ = tr.saved[name]
            xf = nchw(xin.float().cpu()).clone().requires_grad_(True)
            wt = _bf(torch.from_numpy(weights[name]["kernel"])).permute(3, 2, 0, 1).clone().requires_grad_(True)
            F.conv_transpose2d(xf, wt, stride=2, padding=1).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), _masked(tr, name, xin, nhwc(xf.grad)))))
            out.append((name, "dw", _rel(torch.from_numpy(gw[name]["kernel"]), wt.grad.permute(2, 3, 1, 0))))
        elif l.cls == "BatchNormalization":
            xf = tr.saved[name].float().cpu().clone().requires_grad_(True)
            gam = (torch.from_numpy(weights[name]["gamma"]).clone().requires_grad_(True)
                   if "gamma" in weights[name] else None)
            bet = torch.from_numpy(weights[name]["beta"]).clone().requires_grad_(True)
            mean = xf.mean(dim=(0, 1, 2))
            var = ((xf - mean) ** 2).mean(dim=(0, 1, 2))
            y = (xf - mean) * torch.rsqrt(var + l.config["epsilon"])
            if gam is not None:
                y = y * gam
            torch.relu(y + bet).backward(dyf)
            out.append((name, "dx", _rel(dx.float().cpu(), xf.grad)))
            if gam is not None:
                out.append((name, "dgamma", _rel(torch.from_numpy(gw[name]["gamma"]), gam.grad)))
            out.append((name, "dbeta", _rel(torch.from_numpy(gw[name]["beta"]), bet.grad)))
        elif l.cls == "MaxPooling2D":
            xf = nchw(tr.saved[name].float().cpu()).clone().requires_grad_(True)
            if l.config["pool_size"] == 2:
                F.max_pool2d(xf, 2, 2).backward(nchw(dyf))
            else:
                F.max_pool2d(F.pad(xf, (1, 1, 1, 1)), 3, 2).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(xf.grad))))
        elif l.cls == "AveragePooling2D":
            xf = nchw(tr.acts[l.inbound[0]].float().cpu()).clone().requires_grad_(True)
            F.avg_pool2d(xf, 2, 2).backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(xf.grad))))
        elif l.cls == "UpSampling2D":
            xf = nchw(tr.acts[l.inbound[0]].float().cpu()).clone().requires_grad_(True)
            if l.config["interpolation"] == "bilinear":
                y = F.interpolate(xf, scale_factor=2, mode="bilinear", align_corners=False)
            else:
                y = F.interpolate(xf, scale_factor=l.config["size"], mode="nearest")
            y.backward(nchw(dyf))
            out.append((name, "dx", _rel(dx.float().cpu(), nhwc(xf.grad))))
    return out


def accumulation_errors(tr):
    """Every gradient a layer received must be the sum of one contribution per consumer."""
    g = tr.g

    def real_consumers(name):
        n = 0
        for c in tr.consumers[name]:
            cl = g.layers[c]
            if cl.cls == "ZeroPadding2D":
                n += real_consumers(c)
            elif cl.cls in ("Activation", "TemperatureScaledSoftmax") and cl.config.get("activation") != "relu":
                continue
            else:
                n += 1
        return n
    out = []
    for name, parts in tr.accum_log.items():
        want = real_consumers(name)
        total = parts[0].float().cpu()
        for p in parts[1:]:
            total = total + p.float().cpu()
        seen = tr.accum_final[name].float().cpu()
        out.append((name, "fanout %d/%d" % (len(parts), want), _rel(seen, total) if len(parts) == want else 1.0))
    return out


def run(encoder, decoder, h, w, n, classes=3, seed=0):
    """One forward/backward of the HIP trainer on random data; returns (trainer, weights, x, gt)."""
    import miseg_amd  # noqa: F401
    from miseg_amd import netspec, trainer
    from oracle import nets
    spec = nets.build_spec(encoder, decoder, classes)
    weights = nets.init_weights(spec, seed, True)
    rng = np.random.default_rng(seed + 1)
    x = rng.random((n, h, w, 3), dtype=np.float32)
    gt = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, (n, h, w))]
    g = netspec.build_graph(encoder, decoder, (h, w, 3), classes, "softmax", None)
    tr = trainer.Trainer(g, weights, n)
    tr.keep_grads = True
    tr.forward(torch.from_numpy(x).cuda())
    tr.loss_and_metrics(torch.from_numpy(gt).cuda())
    tr.backward()
    torch.cuda.synchronize()
    return tr, weights, x, gt
