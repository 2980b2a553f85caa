"""The torch-CPU op set of oracle/ops_torch.py with every stored tensor rounded to bfloat16
(round-to-nearest-even; the gradient passes straight through the rounding), i.e. an fp32-math
emulation of a mixed-precision run that keeps activations and convolution weights in bf16 and
accumulates in fp32.  TEST INFRASTRUCTURE: see oracle/__init__.py.  It separates "the device
path differs from the fp32 oracle because it stores bf16" from "the device path is wrong":
the device kernels must agree with this op set much more tightly than with the fp32 one.
"""
import torch

from . import ops_torch as T

name = "torch-bf16"
asarray, to_numpy = T.asarray, T.to_numpy


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def r(x):
    return _Round.apply(x)


def zero_pad(x, p):
    return T.zero_pad(x, p)


def conv2d(x, w, stride=1, pad=0, bias=None):
    y = T.conv2d(r(x), r(T._t(w)), stride=stride, pad=pad, bias=bias)
    # the segmentation head (few output channels) keeps fp32 logits for the fused softmax
    return y if y.shape[-1] <= 8 else r(y)


def conv2d_transpose_4x4s2(x, w):
    return r(T.conv2d_transpose_4x4s2(r(x), r(T._t(w))))


def batchnorm(x, gamma, beta, mean, var, eps):
    return r(T.batchnorm(x, gamma, beta, mean, var, eps))


def batchnorm_train(x, gamma, beta, eps):
    y, m, v = T.batchnorm_train(x, gamma, beta, eps)
    return r(y), m, v


def relu(x):
    return T.relu(x)


def maxpool3x3s2_pad1(x):
    return T.maxpool3x3s2_pad1(x)


def maxpool2x2(x):
    return T.maxpool2x2(x)


def avgpool2x2(x):
    return r(T.avgpool2x2(x))


def upsample_nearest(x, s):
    return T.upsample_nearest(x, s)


def upsample_bilinear2x(x):
    return r(T.upsample_bilinear2x(x))


def concat(xs):
    return T.concat(xs)


def add(a, b):
    return r(a + b)


def softmax(x, temperature=1.0):
    return T.softmax(x, temperature)


def sigmoid(x):
    return T.sigmoid(x)
