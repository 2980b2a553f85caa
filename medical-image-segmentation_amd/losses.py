"""Loss and metric objects with the names the reference registers in
CUSTOM_OBJECTS / METRICS (/root/reference utils/model.py:91-106) and selects at
model_train.py:172-179.  Each object is callable as f(gt, pr) on NHWC arrays or
tensors and returns a scalar wrapper with `.numpy()`; the per-class sums are
computed on the GPU by `miseg_loss_sums` (one pass over both tensors), the
handful of scalar operations that follow run on the host.
"""
import numpy as np
import torch

from . import lib as L

SMOOTH = 1e-5


class Scalar(float):
    """A float that also answers `.numpy()` like an eager tf scalar."""

    def numpy(self):
        return np.float32(self)


FOCAL_ALPHA, FOCAL_GAMMA = 0.25, 2.0      # segmentation_models CategoricalFocalLoss defaults


def class_sums(gt, pr, focal=False):
    """(4, C) float64: sum(gt*pr), sum(pr), sum(gt), sum(gt*log(clip(pr/sum_c pr)))
    over all pixels, from one fused GPU pass; with `focal` a fifth row,
    sum(gt * alpha * (1 - q)^gamma * log q), q = clip(pr)."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise L.MisegError("loss/metric sums run on the GPU; no GPU is visible")
    dev = torch.device("cuda", torch.cuda.current_device())

    def prep(a):
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
        return a.to(dev, torch.float32).contiguous()

    g, p = prep(gt), prep(pr)
    if g.shape != p.shape:
        raise ValueError("gt and pr shapes differ: %r vs %r" % (tuple(g.shape), tuple(p.shape)))
    c = g.shape[-1]
    pixels = g.numel() // c
    rows = 5 if focal else 4
    sums = torch.empty(rows * c, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if focal:
        L.check(lib.miseg_loss_sums_focal(g.data_ptr(), p.data_ptr(), sums.data_ptr(), pixels, c,
                                          FOCAL_ALPHA, FOCAL_GAMMA, st))
    else:
        L.check(lib.miseg_loss_sums(g.data_ptr(), p.data_ptr(), sums.data_ptr(), pixels, c, st))
    return sums.cpu().numpy().reshape(rows, c), pixels


def _f1(s):
    tp, spr, sgt = s[0], s[1], s[2]
    return float(np.mean((2 * tp + SMOOTH) / (spr + sgt + SMOOTH)))


def _iou(s):
    tp, spr, sgt = s[0], s[1], s[2]
    return float(np.mean((tp + SMOOTH) / (spr + sgt - tp + SMOOTH)))


def _cce(s, pixels):
    return float(-s[3].sum() / (pixels * s.shape[1]))


def _focal(s, pixels):
    if s.shape[0] < 5:
        raise ValueError("focal loss needs the five-row sums (class_sums(..., focal=True))")
    return float(-s[4].sum() / (pixels * s.shape[1]))


class _Named:
    needs_focal = False

    def __init__(self, name, fn):
        self.name = self.__name__ = name
        self._fn = fn

    def __call__(self, gt, pr):
        s, pixels = class_sums(gt, pr, focal=self.needs_focal)
        return Scalar(self._fn(s, pixels))

    def from_sums(self, s, pixels):
        return self._fn(s, pixels)

    def __repr__(self):
        return "<%s>" % self.name


f1_score = _Named("f1-score", lambda s, n: _f1(s))
iou_score = _Named("iou_score", lambda s, n: _iou(s))
dice_loss = _Named("dice_loss", lambda s, n: 1.0 - _f1(s))
categorical_crossentropy = _Named("categorical_crossentropy", _cce)
cce_dice_loss = _Named("categorical_crossentropy_plus_dice_loss", lambda s, n: _cce(s, n) + 1.0 - _f1(s))


class _Focal(_Named):
    needs_focal = True


categorical_focal_loss = _Focal("focal_loss", _focal)
categorical_focal_dice_loss = _Focal("focal_loss_plus_dice_loss", lambda s, n: _focal(s, n) + 1.0 - _f1(s))

# loss name -> miseg_loss_grad_ex flags
LOSS_FLAGS = {
    "categorical_crossentropy_plus_dice_loss": L.LOSS_CCE | L.LOSS_DICE, "cce_dice": L.LOSS_CCE | L.LOSS_DICE,
    "categorical_crossentropy": L.LOSS_CCE, "cce": L.LOSS_CCE,
    "dice_loss": L.LOSS_DICE,
    "focal_loss": L.LOSS_FOCAL, "focal_loss_plus_dice_loss": L.LOSS_FOCAL | L.LOSS_DICE,
}
BY_NAME = {o.name: o for o in (dice_loss, categorical_crossentropy, cce_dice_loss, categorical_focal_loss,
                               categorical_focal_dice_loss)}
