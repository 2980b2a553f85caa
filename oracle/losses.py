"""numpy restatement of the segmentation_models 1.0.1 losses and metrics that
the reference selects at model_train.py:172-179 and registers at
utils/model.py:91-106 (`sm.losses.cce_dice_loss`,
`categorical_focal_dice_loss`, `categorical_crossentropy`,
`categorical_focal_loss`, `sm.metrics.f1_score`, `iou_score`).  TEST
INFRASTRUCTURE: see oracle/__init__.py.

gt and pr are (N,H,W,C); reductions use float64 accumulators so the oracle's
own rounding stays far below the tolerance of the parity tests.
"""
import numpy as np

SMOOTH = 1e-5
EPS = 1e-7          # keras.backend.epsilon()


def _sums(gt, pr):
    gt = np.asarray(gt, np.float64)
    pr = np.asarray(pr, np.float64)
    ax = (0, 1, 2)
    return (gt * pr).sum(ax), pr.sum(ax), gt.sum(ax)


def f1_score(gt, pr):
    """FScore(beta=1, threshold=None, per_image=False), name 'f1-score'."""
    tp, spr, sgt = _sums(gt, pr)
    fp, fn = spr - tp, sgt - tp
    return float(np.mean((2 * tp + SMOOTH) / (2 * tp + fn + fp + SMOOTH)))


def iou_score(gt, pr):
    """IOUScore(threshold=None, per_image=False), name 'iou_score'."""
    inter, spr, sgt = _sums(gt, pr)
    union = sgt + spr - inter
    return float(np.mean((inter + SMOOTH) / (union + SMOOTH)))


def dice_loss(gt, pr):
    return 1.0 - f1_score(gt, pr)


def categorical_crossentropy(gt, pr):
    """sm functional.categorical_crossentropy: renormalise over classes, clip
    to [eps, 1-eps], mean over ALL elements of -gt*log(pr)."""
    gt = np.asarray(gt, np.float64)
    pr = np.asarray(pr, np.float64)
    pr = pr / pr.sum(axis=-1, keepdims=True)
    pr = np.clip(pr, EPS, 1 - EPS)
    return float(np.mean(-gt * np.log(pr)))


def categorical_focal_loss(gt, pr, alpha=0.25, gamma=2.0):
    gt = np.asarray(gt, np.float64)
    pr = np.clip(np.asarray(pr, np.float64), EPS, 1 - EPS)
    return float(np.mean(-gt * alpha * (1 - pr) ** gamma * np.log(pr)))


def cce_dice_loss(gt, pr):
    """name 'categorical_crossentropy_plus_dice_loss' (utils/model.py:93)."""
    return categorical_crossentropy(gt, pr) + dice_loss(gt, pr)


def categorical_focal_dice_loss(gt, pr):
    """name 'focal_loss_plus_dice_loss' (utils/model.py:94)."""
    return categorical_focal_loss(gt, pr) + dice_loss(gt, pr)


def cce_dice_grad(gt, pr):
    """d(cce_dice_loss)/d(pr), analytic, for the clipped-but-interior case."""
    gt = np.asarray(gt, np.float64)
    pr = np.asarray(pr, np.float64)
    n_el = gt.size
    c = gt.shape[-1]
    s = pr.sum(axis=-1, keepdims=True)
    q = pr / s
    inside = ((q > EPS) & (q < 1 - EPS)).astype(np.float64)
    # L = -1/n * sum gt_c log(q_c),  q_c = pr_c / s
    gq = -gt / np.clip(q, EPS, 1 - EPS) * inside / n_el
    g_cce = gq / s - (gq * pr).sum(axis=-1, keepdims=True) / (s * s)
    tp, spr, sgt = _sums(gt, pr)
    num = 2 * tp + SMOOTH
    den = sgt + spr + SMOOTH            # 2tp + fn + fp = sgt + spr
    g_dice = -(2 * gt * den - num) / (den * den) / c
    return (g_cce + g_dice).astype(np.float64)
