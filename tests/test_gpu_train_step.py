"""Training step of the HIP path (forward on batch statistics, backward, Adam) against torch
autograd and the CPU oracle (oracle/train.py restates model_train.py:284-329).

Tolerances.  Activations, activation gradients and convolution weights are stored in bf16, so
 * layer by layer, on identical inputs: forward outputs and data gradients agree with fp32 torch
   to bf16 rounding (relative L2 <= 6e-3), parameter gradients (fp32 accumulation) to 2e-3;
 * end to end against the fp32 oracle only loss and head gradients are tight.  The random-init,
   batch-normalised network in training mode is ill-conditioned as a function: in the FP32 oracle
   itself a 1e-6 perturbation of the input moves the encoder's parameter gradients by 2-3 %
   (amplification ~10^4), so ANY run that stores activations in bf16 (relative rounding 4e-3) lands
   tens of per cent away there: the oracle's own bf16 emulation (oracle/ops_bf16.py, same rounding
   points as the device path) differs from the fp32 oracle by 0.1 (decoder_stage4b) to 0.69 (encoder).
   The end-to-end gate is therefore relative to that emulation: per parameter tensor, the device
   path may be no further from the fp32 oracle than 1.5x the emulation's own distance (+ 1e-2).
   The layer-wise test is the absolute correctness gate.
PARITY UNPINNED: the reference holds no training fixtures; see oracle/train.py.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONFIGS = [
    ("resnet18", "U-Net", 64, 96, 2),
    ("resnet18", "Linknet", 64, 64, 2),
    ("resnet18", "FPN", 64, 96, 1),
    ("resnet34", "U-Net", 96, 64, 1),
    ("densenet121", "U-Net", 64, 64, 2),
    ("densenet121", "Linknet", 64, 96, 1),
    ("densenet121", "FPN", 64, 64, 1),
    ("densenet169", "FPN", 64, 64, 1),          # BASELINE config 5's network
    ("densenet169", "Linknet", 64, 64, 1),      # the reference's preferred model
    ("resnet50", "U-Net", 64, 64, 1),           # bottleneck units
    ("vgg16", "U-Net", 64, 64, 1),              # biased convolutions with ReLU, 2x2 max pools, centre blocks
    ("vgg16", "Linknet", 64, 64, 2),
    ("vgg19", "FPN", 64, 64, 1),
]


@pytest.mark.parametrize("encoder,decoder,h,w,n", CONFIGS)
def test_layerwise_forward_backward(pkg, encoder, decoder, h, w, n):
    import train_selfcheck as sc
    tr, weights, x, gt = sc.run(encoder, decoder, h, w, n)
    bad = []
    for name, kind, err in sc.forward_errors(tr, weights):
        # only rounding flips may differ; "fwd-plain": a layer the trainer runs in a split form whose bf16
        # weights are rounded after a sum of taps, compared with the layer as the graph states it
        if not err <= (5e-3 if kind == "fwd-plain" else 2e-3):
            bad.append((name, kind, err))
    for name, kind, err in sc.backward_errors(tr, weights, x):
        tol = 6e-3 if kind == "dx" else 2e-3
        if not err <= tol:
            bad.append((name, kind, err))
    acc = sc.accumulation_errors(tr)
    assert len(acc) > 10
    for name, kind, err in acc:
        if not err <= 6e-3:
            bad.append((name, kind, err))
    assert not bad, bad[:20]
    # the two full-resolution BatchNorm + ReLU pairs at the end of a U-Net decoder are applied by the layers that
    # read them (forward convolution and its weight gradient): no apply pass, no activated tensor
    assert len(tr.fwd_prologues) == (2 if decoder == "U-Net" else 0), tr.fwd_prologues


def test_loss_and_head_gradients_vs_oracle(pkg):
    import train_selfcheck as sc
    from oracle import train, losses
    tr, weights, x, gt = sc.run("resnet18", "U-Net", 128, 192, 2)
    loss, probs, grads, stats = train.loss_and_grads("resnet18", "U-Net", weights, x, gt)
    s = tr.loss_sums.cpu().numpy().reshape(4, 3)
    tp, spr, sgt, ce = s
    mine = -ce.sum() / gt.size + 1.0 - np.mean((2 * tp + 1e-5) / (sgt + spr + 1e-5))
    assert abs(mine - loss) <= 1e-2 * abs(loss), (mine, loss)
    gw = sc.grads_as_weights(tr)
    for layer, wname, tol in (("final_conv", "kernel", 2e-2), ("final_conv", "bias", 2e-2),
                              ("decoder_stage4b_bn", "gamma", 3e-2), ("decoder_stage4b_bn", "beta", 3e-2)):
        ref = grads[layer][wname]
        err = np.linalg.norm(gw[layer][wname] - ref) / np.linalg.norm(ref)
        assert err <= tol, (layer, wname, err)
    # the whole gradient still points the same way
    a = np.concatenate([gw[l][n_].ravel() for l in grads for n_ in grads[l]])
    b = np.concatenate([grads[l][n_].ravel() for l in grads for n_ in grads[l]])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.8, cos
    # batch statistics and the moving averages they feed
    for name in ("bn0", "stage2_unit1_bn1", "decoder_stage4b_bn"):
        m, v = stats[name]
        st = tr.bn_state[name]
        assert np.allclose(st["mean"][:m.size].cpu().numpy(), m, rtol=0.1, atol=0.05 * np.sqrt(v).max())
        assert np.allclose(st["var"][:m.size].cpu().numpy(), v, rtol=0.1)
        count = tr.saved[name].pixels
        mm, mv = train.moving_update(weights[name]["moving_mean"], weights[name]["moving_variance"],
                                     st["mean"][:m.size].cpu().numpy(), st["var"][:m.size].cpu().numpy(), count)
        assert np.allclose(tr.moving[name][0].cpu().numpy(), mm, rtol=1e-4, atol=1e-5)
        assert np.allclose(tr.moving[name][1].cpu().numpy(), mv, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("encoder,decoder,h,w,n", [("resnet18", "U-Net", 128, 192, 2), ("densenet121", "FPN", 64, 64, 2),
                                                  ("vgg16", "U-Net", 64, 96, 2)])
def test_gradients_no_further_from_fp32_than_the_bf16_emulation(pkg, encoder, decoder, h, w, n):
    """End to end: every parameter gradient of one HIP training step against the fp32 oracle,
    with the oracle's bf16-storage emulation as the yardstick (see the module docstring)."""
    import train_selfcheck as sc
    from oracle import train, ops_bf16
    tr, weights, x, gt = sc.run(encoder, decoder, h, w, n)
    gw = sc.grads_as_weights(tr)
    loss32, _, g32, _ = train.loss_and_grads(encoder, decoder, weights, x, gt)
    loss16, _, g16, _ = train.loss_and_grads(encoder, decoder, weights, x, gt, ops=ops_bf16)
    s = tr.loss_sums.cpu().numpy().reshape(4, 3)
    tp, spr, sgt, ce = s
    mine = -ce.sum() / gt.size + 1.0 - np.mean((2 * tp + 1e-5) / (sgt + spr + 1e-5))
    assert abs(mine - loss32) <= max(2 * abs(loss16 - loss32), 1e-3 * abs(loss32)), (mine, loss32, loss16)
    bad, ratios = [], []
    for l in g32:
        for k in g32[l]:
            ref = g32[l][k]
            nr = np.linalg.norm(ref)
            if nr == 0:
                continue
            e_dev = np.linalg.norm(gw[l][k] - ref) / nr
            e_emu = np.linalg.norm(g16[l][k] - ref) / nr
            ratios.append(e_dev / max(e_emu, 1e-3))
            # a tensor of a few elements at the far end of the network (bn_data's beta: 3 values whose
            # emulation error is already tens of per cent) is two draws of the same amplified rounding
            # noise: its ratio scatters more than that of a tensor with thousands of elements
            bar = (2.5 if ref.size < 8 else 1.5) * e_emu + 1e-2
            if e_emu > 0.7:
                # the emulation's own gradient of this tensor is mostly amplified rounding noise already; two
                # independent draws of such noise around the same signal are up to sqrt(2) of its norm apart
                bar = max(bar, 1.45)
            if e_dev > bar:
                bad.append((l, k, float(e_dev), float(e_emu)))
    assert not bad, bad[:10]
    assert np.median(ratios) < 1.2, np.median(ratios)      # on the whole, as close as the emulation
    a = np.concatenate([gw[l][k].ravel() for l in g32 for k in g32[l]])
    b16 = np.concatenate([g16[l][k].ravel() for l in g32 for k in g32[l]])
    b32 = np.concatenate([g32[l][k].ravel() for l in g32 for k in g32[l]])
    cos = lambda u, v: float(u @ v / (np.linalg.norm(u) * np.linalg.norm(v)))
    assert cos(a, b32) >= cos(b16, b32) - 0.03, (cos(a, b32), cos(b16, b32))


def test_training_reduces_loss_like_the_oracle(pkg):
    """30 Adam steps on one fixed batch: the HIP path and the fp32 CPU oracle both fit it."""
    import torch
    from miseg_amd import netspec, trainer
    from oracle import nets, train
    enc, dec, h, w, n, c = "resnet18", "U-Net", 64, 96, 2, 3
    weights = nets.init_weights(nets.build_spec(enc, dec, c), 3, False)
    rng = np.random.default_rng(5)
    x = rng.random((n, h, w, 3), dtype=np.float32)
    lab = (x[..., 0] > 0.5).astype(np.int64) + (x[..., 1] > 0.7).astype(np.int64)
    gt = np.eye(c, dtype=np.float32)[lab]
    g = netspec.build_graph(enc, dec, (h, w, 3), c, "softmax", None)
    tr = trainer.Trainer(g, weights, n, learning_rate=1e-3)
    xd, gd = torch.from_numpy(x).cuda(), torch.from_numpy(gt).cuda()
    mine = []
    for _ in range(30):
        sums, _ = tr.train_step(xd, gd)
        tp, spr, sgt, ce = sums.cpu().numpy().reshape(4, c)
        mine.append(-ce.sum() / gt.size + 1.0 - np.mean((2 * tp + 1e-5) / (sgt + spr + 1e-5)))
    # oracle: same data, same initial weights, Keras Adam in numpy
    wts = {k: {n_: v.copy() for n_, v in d.items()} for k, d in weights.items()}
    m = {k: {n_: np.zeros_like(v) for n_, v in d.items()} for k, d in wts.items()}
    v2 = {k: {n_: np.zeros_like(v) for n_, v in d.items()} for k, d in wts.items()}
    ref = []
    for t in range(1, 31):
        loss, _, grads, stats = train.loss_and_grads(enc, dec, wts, x, gt)
        ref.append(loss)
        for l in grads:
            for n_ in grads[l]:
                wts[l][n_], m[l][n_], v2[l][n_] = train.adam_update(wts[l][n_], grads[l][n_], m[l][n_], v2[l][n_], t)
    assert abs(mine[0] - ref[0]) <= 1e-2 * ref[0], (mine[0], ref[0])
    assert mine[-1] < 0.75 * mine[0], mine
    assert abs(mine[-1] - ref[-1]) <= 0.15 * ref[-1], (mine[-1], ref[-1])


@pytest.mark.parametrize("classes,loss", [(3, "focal_loss_plus_dice_loss"), (3, "focal_loss"), (1, "dice_loss")])
def test_focal_loss_and_sigmoid_head_train(pkg, classes, loss):
    """The remaining selectable losses (model_train.py:172-177) and the single-class sigmoid head
    (utils/model.py:180) go through a few optimisation steps: the reported loss equals the numpy
    restatement (oracle/losses.py) on the probabilities the step produced, and it decreases."""
    import torch
    from miseg_amd import netspec, trainer, losses as LS
    from oracle import nets, losses as OL
    act = "sigmoid" if classes == 1 else "softmax"
    w = nets.init_weights(nets.build_spec("resnet18", "U-Net", classes), 0, True)
    g = netspec.build_graph("resnet18", "U-Net", (64, 96, 3), classes, act, None)
    rng = np.random.default_rng(3)
    x = rng.random((2, 64, 96, 3), dtype=np.float32)
    if classes == 1:
        y = (rng.random((2, 64, 96, 1)) < 0.4).astype(np.float32)
    else:
        y = np.eye(classes, dtype=np.float32)[np.repeat(np.repeat(rng.integers(0, classes, (2, 8, 12)), 8, 1), 8, 2)]
    tr = trainer.Trainer(g, w, 2, learning_rate=1e-3, loss=loss)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    vals = []
    obj = LS.BY_NAME[loss]
    for it in range(12):
        sums, pixels = tr.train_step(xd, yd)
        s = sums.cpu().numpy().reshape(-1, classes)
        vals.append(obj.from_sums(s, pixels))
        if it == 0:
            pr = tr.probs.cpu().numpy()
            want = {"focal_loss_plus_dice_loss": OL.categorical_focal_dice_loss, "focal_loss": OL.categorical_focal_loss,
                    "dice_loss": OL.dice_loss}[loss](y, pr)
            assert abs(vals[0] - want) <= 1e-4 * max(1.0, abs(want)), (vals[0], want)
    assert np.isfinite(vals).all()
    assert vals[-1] < vals[0], vals
