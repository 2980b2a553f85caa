"""CPU restatement of one optimisation step of the reference's training loop
(`model.fit` at /root/reference model_train.py:284-329 with the loss of :172-179 and
Adam(learning_rate) of utils/model.py:234), on torch-CPU autograd over the oracle's own
forward graphs.  TEST INFRASTRUCTURE: see oracle/__init__.py.

PARITY UNPINNED: the reference ships no training fixtures (no loss curves, no
checkpoints); the step is pinned only by the published Keras/TensorFlow semantics restated
here: BatchNormalization on batch statistics with moving = m*moving + (1-m)*batch (unbiased
variance, as the fused TF kernel does), segmentation_models cce_dice_loss, and
tf.keras Adam (epsilon outside the square root, 1e-7).
"""
from collections import OrderedDict

import numpy as np
import torch

from . import nets, ops_torch

SMOOTH = 1e-5
EPS = 1e-7


def cce_dice_loss_torch(gt, pr, use_cce=True, use_dice=True):
    """segmentation_models cce_dice_loss on torch tensors (float64 inside)."""
    gt = gt.double()
    pr = pr.double()
    loss = 0.0
    if use_cce:
        q = pr / pr.sum(dim=-1, keepdim=True)
        q = torch.clamp(q, EPS, 1 - EPS)
        loss = loss + torch.mean(-gt * torch.log(q))
    if use_dice:
        ax = (0, 1, 2)
        tp = (gt * pr).sum(ax)
        fp = pr.sum(ax) - tp
        fn = gt.sum(ax) - tp
        f = (2 * tp + SMOOTH) / (2 * tp + fn + fp + SMOOTH)
        loss = loss + 1.0 - f.mean()
    return loss


def loss_and_grads(encoder, decoder, weights, x, gt, activation="softmax", temperature=1.0,
                   use_cce=True, use_dice=True, threads=None, ops=None):
    """Training-mode forward + backward.  weights: {layer: {name: ndarray}} (Keras layouts).
    Returns (loss, probs ndarray, grads {layer: {name: ndarray}}, batch_stats)."""
    if threads:
        torch.set_num_threads(threads)
    tw = OrderedDict()
    leaves = []
    for lname, d in weights.items():
        tw[lname] = OrderedDict()
        for n, v in d.items():
            t = torch.from_numpy(np.array(v, dtype=np.float32))
            if n not in ("moving_mean", "moving_variance"):
                t.requires_grad_(True)
                leaves.append((lname, n, t))
            tw[lname][n] = t
    stats = OrderedDict()
    probs = nets.forward(ops or ops_torch, encoder, decoder, tw, torch.from_numpy(np.asarray(x, np.float32)),
                         activation=activation, temperature=temperature, training=True,
                         batch_stats=stats)
    loss = cce_dice_loss_torch(torch.from_numpy(np.asarray(gt, np.float32)), probs, use_cce, use_dice)
    loss.backward()
    grads = OrderedDict()
    for lname, n, t in leaves:
        g = t.grad
        grads.setdefault(lname, OrderedDict())[n] = (np.zeros(t.shape, np.float32) if g is None
                                                     else g.numpy().copy())
    bs = OrderedDict((k, (m.detach().numpy().copy(), v.detach().numpy().copy())) for k, (m, v) in stats.items())
    return float(loss.detach()), probs.detach().numpy(), grads, bs


def moving_update(moving_mean, moving_var, mean, var, count, momentum=0.99):
    """Keras/TF fused batch norm side effect: unbiased variance enters the moving average."""
    unb = var * (count / max(count - 1.0, 1.0))
    return (momentum * moving_mean + (1 - momentum) * mean,
            momentum * moving_var + (1 - momentum) * unb)


def adam_update(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam, step t (1-based); returns (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
