"""The fused DenseNet layer kernel (miseg_dense_layer_fwd: BatchNorm+ReLU -> 1x1 -> BatchNorm+ReLU ->
3x3, appended to the concatenation buffer; keras_applications `conv_block`, reached from
/root/reference utils/model.py:188-211) against the CPU oracle and against the two separate
miseg_conv2d_fwd launches it replaces, on both storage formats, through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FORMATS = {"bf16": (torch.bfloat16, 0), "fp16": (torch.float16, 1)}


@pytest.fixture(scope="module")
def env():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    L.check(lib.miseg_device_check())
    return L, lib, torch.device("cuda:0")


def rnd(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dt).float().numpy()


def oracle_layer(x, s0, t0, w1, s1, t1, w2, dt):
    """fp32 arithmetic on the stored operands, re-rounding where the kernels store 16-bit values."""
    from oracle import ops_torch as T
    a = np.maximum(x * s0 + t0, 0)
    a = rnd(a, dt)
    m = T.to_numpy(T.conv2d(T.asarray(a), w1, stride=1, pad=0))
    m = rnd(np.maximum(m * s1 + t1, 0), dt)
    return T.to_numpy(T.conv2d(T.asarray(m), w2, stride=1, pad=1))


def descs(L, fmt, cat, c, ctot, n, h, w, w1, w2, s0, t0, s1, t1, mid=None):
    d1, d2 = L.ConvDesc(), L.ConvDesc()
    for d in (d1, d2):
        d.fmt = fmt
        d.N, d.H, d.W, d.Ho, d.Wo = n, h, w, h, w
        d.stride_y = d.stride_x = 1
        d.temperature = 1.0
        d.out_dtype = L.DT_BF16
    d1.inp, d1.weight = cat.data_ptr(), w1.data_ptr()
    d1.Cin, d1.Cout, d1.KH, d1.KW = c, 128, 1, 1
    d1.in_pix_stride, d1.in_row_stride, d1.in_img_stride = ctot, ctot * w, ctot * w * h
    d1.in_scale, d1.in_shift = s0.data_ptr(), t0.data_ptr()
    d1.scale, d1.shift, d1.act = s1.data_ptr(), t1.data_ptr(), L.ACT_RELU
    d2.weight = w2.data_ptr()
    d2.Cin, d2.Cout, d2.KH, d2.KW, d2.pad_y, d2.pad_x = 128, 32, 3, 3, 1, 1
    d2.out = cat.data_ptr() + 2 * c
    d2.out_pix_stride, d2.out_row_stride, d2.out_img_stride = ctot, ctot * w, ctot * w * h
    if mid is not None:      # the two-launch form: 1x1 -> mid -> 3x3
        d1.out = mid.data_ptr()
        d1.out_pix_stride, d1.out_row_stride, d1.out_img_stride = 128, 128 * w, 128 * w * h
        d2.inp = mid.data_ptr()
        d2.in_pix_stride, d2.in_row_stride, d2.in_img_stride = 128, 128 * w, 128 * w * h
    return d1, d2


CASES = [
    # n, h, w, C
    (1, 8, 16, 64),        # exactly one patch, two chunks
    (2, 11, 37, 96),       # ragged patches on both edges, odd chunk count
    (1, 3, 5, 160),        # smaller than a patch (the deepest block at test sizes)
    (1, 30, 40, 32),       # a single chunk
    (2, 16, 48, 352),      # eleven chunks: the ring wraps several times
    (1, 9, 17, 1632),      # the widest layer of DenseNet-169
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("storage", ["bf16", "fp16"])
def test_dense_layer(env, case, storage):
    L, lib, dev = env
    dt, fmt = FORMATS[storage]
    n, h, w, c = case
    rng = np.random.default_rng(abs(hash(case)) % 2**32)
    ctot = c + 64                      # the layer's 32 channels and 32 more that must stay untouched
    x = rnd(rng.normal(0, 1, (n, h, w, c)), dt)
    w1 = rnd(rng.normal(0, 1.0 / np.sqrt(c), (1, 1, c, 128)), dt)
    w2 = rnd(rng.normal(0, 0.03, (3, 3, 128, 32)), dt)
    s0 = rng.uniform(0.5, 1.5, c).astype(np.float32)
    t0 = rng.normal(0, 0.5, c).astype(np.float32)
    s1 = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    t1 = rng.normal(0, 0.5, 128).astype(np.float32)
    want = oracle_layer(x, s0, t0, w1, s1, t1, w2, dt)

    def cat_buffer():
        t = torch.full((n, h, w, ctot), 7.0, dtype=dt, device=dev)
        t[..., :c] = torch.from_numpy(x).to(dt).to(dev)
        return t

    to = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    w1d = to(w1.transpose(3, 0, 1, 2)).to(dt)
    w2d = to(w2.transpose(3, 0, 1, 2)).to(dt)
    s0d, t0d, s1d, t1d = to(s0), to(t0), to(s1), to(t1)

    cat = cat_buffer()
    d1, d2 = descs(L, fmt, cat, c, ctot, n, h, w, w1d, w2d, s0d, t0d, s1d, t1d)
    L.check(lib.miseg_dense_layer_fwd(C.byref(d1), C.byref(d2), None))
    torch.cuda.synchronize()
    got = cat[..., c:c + 32].float().cpu().numpy()
    assert torch.equal(cat[..., :c].float().cpu(), torch.from_numpy(x)), "the layer's input was modified"
    assert torch.all(cat[..., c + 32:].float() == 7.0), "wrote past its 32 channels"

    # against the oracle: fp32 accumulation order + one 16-bit rounding of the result; with fp16
    # storage the input BatchNorm is evaluated in fp16 (coefficients and result rounded to 11 bits)
    tol = 2e-2 if storage == "bf16" else 4e-3
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= tol * scale, (np.abs(got - want).max(), scale)

    # against the two launches it replaces
    cat2 = cat_buffer()
    mid = torch.zeros((n, h, w, 128), dtype=dt, device=dev)
    e1, e2 = descs(L, fmt, cat2, c, ctot, n, h, w, w1d, w2d, s0d, t0d, s1d, t1d, mid=mid)
    L.check(lib.miseg_conv2d_fwd(C.byref(e1), None))
    L.check(lib.miseg_conv2d_fwd(C.byref(e2), None))
    torch.cuda.synchronize()
    two = cat2[..., c:c + 32].float().cpu().numpy()
    assert np.abs(got - two).max() <= tol * scale
    # what differs is a rounding flip of an operand or of the result: with bf16 most values are
    # identical; with fp16 the input BatchNorm's own rounding differs (fp16 arithmetic here, fp32
    # there), so most results differ in their last bit and the mean difference is what stays small
    if storage == "bf16":
        assert np.mean(got != two) < 0.05, np.mean(got != two)
    assert np.abs(got - two).mean() <= tol * scale / 20, (np.abs(got - two).mean(), scale)


def test_dense_layer_rejects(env):
    L, lib, dev = env
    cat = torch.zeros((1, 8, 16, 128), dtype=torch.bfloat16, device=dev)
    w1 = torch.zeros((128, 64), dtype=torch.bfloat16, device=dev)
    w2 = torch.zeros((32, 3, 3, 128), dtype=torch.bfloat16, device=dev)
    v = torch.zeros(128, dtype=torch.float32, device=dev)
    d1, d2 = descs(L, 0, cat, 64, 128, 1, 8, 16, w1, w2, v, v, v, v)
    d2.Cout = 64
    assert lib.miseg_dense_layer_fwd(C.byref(d1), C.byref(d2), None) == L.E_UNSUPPORTED_SHAPE
    d2.Cout = 32
    d1.in_scale = None
    assert lib.miseg_dense_layer_fwd(C.byref(d1), C.byref(d2), None) == L.E_INVALID_ARG
