"""The fused U-Net tail (miseg_conv_c32_head_fwd: 3x3 32->32 + scale/shift + ReLU, then the
3x3 32->classes head with softmax / sigmoid and the collapse mask) against (a) the CPU oracle
and (b) the same two layers run as separate miseg_conv2d_fwd calls.  The intermediate tensor is
bf16 in both; differences come from fp32 summation order only (tolerances below)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_kernels import bf16_round, dev_bf16, dev_f32, env, storage, new_desc, h16  # noqa: F401

pytestmark = pytest.mark.gpu


def descs(L, dev, x, wb, sb, tb, wh, bias, classes, act, temperature, keep):
    n, h, w, _ = x.shape
    xin = dev_bf16(x, dev)
    wbt = dev_bf16(wb.transpose(3, 0, 1, 2), dev)
    wht = dev_bf16(wh.transpose(3, 0, 1, 2), dev)
    mid = torch.zeros((n, h, w, 32), dtype=h16(), device=dev)
    probs = torch.zeros((n, h, w, classes), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, h, w, classes), dtype=torch.uint8, device=dev)
    s, t, b = dev_f32(sb, dev), dev_f32(tb, dev), dev_f32(bias, dev)
    keep += [xin, wbt, wht, mid, probs, mask, s, t, b]

    def base(cout):
        d = new_desc(L)
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, 32, h, w, cout
        d.KH = d.KW = 3
        d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = 32, 32 * w, 32 * w * h
        d.temperature = 1.0
        return d

    body = base(32)
    body.inp, body.weight, body.scale, body.shift = xin.data_ptr(), wbt.data_ptr(), s.data_ptr(), t.data_ptr()
    body.act = L.ACT_RELU
    body.out = mid.data_ptr()
    body.out_pix_stride, body.out_row_stride, body.out_img_stride = 32, 32 * w, 32 * w * h
    head = base(classes)
    head.inp, head.weight, head.shift = mid.data_ptr(), wht.data_ptr(), b.data_ptr()
    head.act = act
    head.temperature = temperature
    head.out = probs.data_ptr()
    head.out_dtype = L.DT_F32
    head.out_pix_stride, head.out_row_stride, head.out_img_stride = classes, classes * w, classes * w * h
    head.collapse_out = mask.data_ptr()
    head.collapse_value = 127
    return body, head, mid, probs, mask


@pytest.mark.parametrize("shape", [(1, 8, 32), (1, 12, 20), (2, 17, 33), (1, 40, 70), (1, 9, 100)])
@pytest.mark.parametrize("classes,act_name,temperature", [(3, "softmax", 1.0), (3, "softmax", 0.1),
                                                          (1, "sigmoid", 1.0), (4, "softmax", 1.0)])
def test_fused_tail(env, shape, classes, act_name, temperature):
    from oracle import ops_torch as T, post
    L, lib, dev = env
    n, h, w = shape
    rng = np.random.default_rng(hash((shape, classes, temperature)) % 2**32)
    x = bf16_round(np.maximum(rng.normal(0, 1, (n, h, w, 32)), 0))
    wb = bf16_round(rng.normal(0, 0.08, (3, 3, 32, 32)))
    wh = bf16_round(rng.normal(0, 0.08, (3, 3, 32, classes)))
    sb = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    tb = rng.normal(0, 0.2, 32).astype(np.float32)
    bias = rng.normal(0, 0.1, classes).astype(np.float32)
    act = L.ACT_SOFTMAX if act_name == "softmax" else L.ACT_SIGMOID
    keep = []
    body, head, mid, probs, mask = descs(L, dev, x, wb, sb, tb, wh, bias, classes, act, temperature, keep)

    # (b) the two layers as separate launches
    L.check(lib.miseg_conv2d_fwd(C.byref(body), None))
    L.check(lib.miseg_conv2d_fwd(C.byref(head), None))
    torch.cuda.synchronize()
    sep_p, sep_m = probs.cpu().numpy().copy(), mask.cpu().numpy().copy()
    mid_sep = mid.float().cpu().numpy()
    probs.fill_(-1.0)
    mask.fill_(9)
    body.out = None
    head.inp = None
    L.check(lib.miseg_conv_c32_head_fwd(C.byref(body), C.byref(head), None))
    torch.cuda.synchronize()
    got_p, got_m = probs.cpu().numpy(), mask.cpu().numpy()
    assert np.isfinite(got_p).all() and got_p.min() >= 0.0
    # (an intermediate value whose bf16 rounding flips between the two code paths moves the 27
    # probabilities of its 3x3 neighbourhood by up to ~1e-2; such flips are rare)
    close = np.abs(got_p - sep_p) <= 1e-5
    assert close.mean() >= 0.99, "fused vs separate launches: %.5f of the elements within 1e-5" % close.mean()
    assert np.abs(got_p - sep_p).max() < 2e-2
    assert (got_m == sep_m).all(axis=-1).mean() >= 0.995

    # (a) the oracle, fed the same bf16 operands; the intermediate is rounded to bf16 as on the device
    m = T.to_numpy(T.conv2d(T.asarray(x), wb, stride=1, pad=1)) * sb + tb
    m = bf16_round(np.maximum(m, 0))
    assert np.abs(mid_sep - m).max() <= 2e-2 * max(1.0, np.abs(m).max())
    logits = T.to_numpy(T.conv2d(T.asarray(m), wh, stride=1, pad=1)) + bias
    if act_name == "softmax":
        z = logits / temperature
        e = np.exp(z - z.max(-1, keepdims=True))
        want = e / e.sum(-1, keepdims=True)
    else:
        want = 1.0 / (1.0 + np.exp(-logits))
    # a bf16 rounding flip of one intermediate value moves a logit by ~1e-3 (x 1/temperature)
    assert np.abs(got_p - want).max() < (3e-2 if temperature < 1 else 5e-3)
    if classes > 1:
        np.testing.assert_array_equal(got_m, np.stack([post.collapse_probabilities(p.copy()) for p in got_p]))


def test_fused_tail_rejects_other_shapes(env):
    L, lib, dev = env
    keep = []
    x = np.zeros((1, 8, 32, 32), np.float32)
    body, head, *_ = descs(L, dev, x, np.zeros((3, 3, 32, 32), np.float32), np.ones(32, np.float32),
                           np.zeros(32, np.float32), np.zeros((3, 3, 32, 3), np.float32),
                           np.zeros(3, np.float32), 3, L.ACT_SOFTMAX, 1.0, keep)
    body.Cout = 64
    assert lib.miseg_conv_c32_head_fwd(C.byref(body), C.byref(head), None) == L.E_UNSUPPORTED_SHAPE
    body.Cout = 32
    head.act = L.ACT_RELU
    assert lib.miseg_conv_c32_head_fwd(C.byref(body), C.byref(head), None) == L.E_INVALID_ARG
