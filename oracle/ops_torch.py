"""torch-CPU second opinion for oracle/ops_np.py: the same op set composed from
torch.nn.functional, independent of the numpy restatement.  TEST
INFRASTRUCTURE: see oracle/__init__.py.  This is also the fast CPU variant the
benchmark times as `cpu_baseline` (kind "port").

Tensors are NHWC at the interface (as the reference's tf.keras tensors are,
utils/predict.py:158-159) and NCHW inside each call.
"""
import numpy as np
import torch
import torch.nn.functional as F

name = "torch"


def asarray(x):
    if isinstance(x, torch.Tensor):
        return x.float()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))


def to_numpy(x):
    return x.detach().cpu().numpy()


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v, dtype=np.float32))


def zero_pad(x, p):
    return F.pad(x, (0, 0, p, p, p, p)) if p else x


def conv2d(x, w, stride=1, pad=0, bias=None):
    wt = _t(w).permute(3, 2, 0, 1)  # HWIO -> OIHW
    b = None if bias is None else _t(bias)
    return _nhwc(F.conv2d(_nchw(x), wt, b, stride=stride, padding=pad))


def conv2d_transpose_4x4s2(x, w):
    wt = _t(w).permute(3, 2, 0, 1)  # (kh,kw,Cout,Cin) -> (Cin,Cout,kh,kw)
    return _nhwc(F.conv_transpose2d(_nchw(x), wt, stride=2, padding=1))


def batchnorm(x, gamma, beta, mean, var, eps):
    g = None if gamma is None else _t(gamma)
    return _nhwc(F.batch_norm(_nchw(x), _t(mean), _t(var), g, _t(beta), False, 0.0, eps))


def batchnorm_train(x, gamma, beta, eps):
    """Batch-statistics normalisation; returns (y, mean, biased variance)."""
    mean = x.mean(dim=(0, 1, 2))
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + eps)
    if gamma is not None:
        y = y * _t(gamma)
    return y + _t(beta), mean, var


def relu(x):
    return F.relu(x)


def maxpool3x3s2_pad1(x):
    return _nhwc(F.max_pool2d(F.pad(_nchw(x), (1, 1, 1, 1)), 3, 2))


def maxpool2x2(x):
    return _nhwc(F.max_pool2d(_nchw(x), 2, 2))


def avgpool2x2(x):
    return _nhwc(F.avg_pool2d(_nchw(x), 2, 2))


def upsample_nearest(x, s):
    return _nhwc(F.interpolate(_nchw(x), scale_factor=s, mode="nearest"))


def upsample_bilinear2x(x):
    return _nhwc(F.interpolate(_nchw(x), scale_factor=2, mode="bilinear", align_corners=False))


def concat(xs):
    return torch.cat(list(xs), dim=-1)


def add(a, b):
    return a + b


def softmax(x, temperature=1.0):
    return torch.softmax(x / temperature, dim=-1)


def sigmoid(x):
    return torch.sigmoid(x)
