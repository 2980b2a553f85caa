"""The fused FPN end (miseg_fpn_head_fwd: bilinear x2 up-sampling + 3x3 head convolution + softmax /
sigmoid + collapse mask; segmentation_models `final_upsampling` + `head_conv`, reached from
/root/reference utils/model.py:200-211) against a torch-CPU restatement, through the C ABI, on both
storage formats, including extents that leave ragged 8 x 32 output patches."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


@pytest.mark.parametrize("shape", [(1, 4, 16, 3), (2, 7, 21, 3), (1, 16, 40, 1), (1, 5, 9, 4)])
@pytest.mark.parametrize("storage", ["bf16", "fp16"])
def test_fpn_head(env, shape, storage):
    L, lib, dev = env
    dt, fmt = FORMATS[storage]
    n, h, w, classes = shape
    rng = np.random.default_rng(h * 100 + w)
    extra = 32                                            # the input is a channel slice of a wider buffer
    x = torch.from_numpy(rng.normal(0, 1, (n, h, w, 128)).astype(np.float32)).to(dt)
    k = torch.from_numpy(rng.normal(0, 0.05, (3, 3, 128, classes)).astype(np.float32)).to(dt)
    bias = torch.from_numpy(rng.normal(0, 0.3, classes).astype(np.float32))
    # restatement: bilinear x2 with half-pixel centres, stored in the 16-bit format, 3x3 head with zero padding
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    up = up.to(dt).float()
    logits = F.conv2d(up, k.float().permute(3, 2, 0, 1), bias=bias, padding=1).permute(0, 2, 3, 1)
    want = torch.sigmoid(logits) if classes == 1 else torch.softmax(logits, dim=-1)

    xin = torch.zeros((n, h, w, 128 + extra), dtype=dt, device=dev)
    xin[..., :128] = x.to(dev)
    wd = torch.zeros((16, 9, 128), dtype=dt, device=dev)
    wd[:classes] = k.permute(3, 0, 1, 2).reshape(classes, 9, 128).to(dev)
    bd = bias.to(dev)
    out = torch.full((n, 2 * h, 2 * w, classes), -1.0, dtype=torch.float32, device=dev)
    mask = torch.full((n, 2 * h, 2 * w, classes), 9, dtype=torch.uint8, device=dev)
    d = L.FpnHeadDesc()
    d.inp, d.wd, d.bias, d.out, d.collapse_out = xin.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), mask.data_ptr()
    d.N, d.H, d.W = n, h, w
    cs = 128 + extra
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cs, cs * w, cs * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = classes, classes * 2 * w, classes * 4 * w * h
    d.classes, d.act, d.temperature, d.collapse_value, d.fmt = classes, (L.ACT_SIGMOID if classes == 1 else L.ACT_SOFTMAX), 1.0, 127, fmt
    L.check(lib.miseg_fpn_head_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = out.cpu()
    # identical 16-bit operands (the interpolation is evaluated in the same order in fp32): what is
    # left is the accumulation order of the head and a few rounding flips of up-sampled values
    assert (got - want).abs().max() < 2e-3, (got - want).abs().max()
    if classes > 1:
        from oracle import post
        m_want = np.stack([post.collapse_probabilities(p) for p in got.numpy()])
        np.testing.assert_array_equal(mask.cpu().numpy(), m_want)


def test_fpn_head_rejects(env):
    L, lib, dev = env
    d = L.FpnHeadDesc()
    assert lib.miseg_fpn_head_fwd(C.byref(d), None) == L.E_INVALID_ARG
