"""miseg_linknet_block_fwd through the C ABI: a LinkNet decoder block (1x1 C -> C/4, Conv2DTranspose(4, 2, 'same')
C/4 -> C/4, 1x1 C/4 -> 64, each BatchNorm + ReLU, then the skip addition; segmentation_models DecoderBlock reached
from /root/reference utils/model.py:188-198) against a numpy restatement with the kernel's rounding points (the two
C/4-channel tensors rounded to the storage format), for both channel counts, ragged 8 x 64 output patches, several
images, and input / skip / output given as channel slices of wider buffers."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_kernels import bf16_round, dev_bf16, dev_f32, env, storage, h16  # noqa: F401

pytestmark = pytest.mark.gpu


def round16(a):
    """Round to the storage format under test (bf16_round follows the `storage` fixture)."""
    return bf16_round(a)


def reference(x, ka, kb, kc, sa, ta, sb, tb, sc, tc, skip):
    """x (N,H,W,C); ka (C,m), kb Keras transpose kernel (4,4,m_out,m_in), kc (m,64); fp32 arithmetic."""
    n, h, w, _ = x.shape
    m = ka.shape[1]
    a = round16(np.maximum(x @ ka * sa + ta, 0))
    # Conv2DTranspose(4, 2, 'same'): out[2y + p, 2x + q] = sum_{dy,dx} in[y - 1 + p + dy, x - 1 + q + dx] K[3 - p - 2dy, 3 - q - 2dx]
    ap = np.zeros((n, h + 2, w + 2, m), np.float32)
    ap[:, 1:-1, 1:-1] = a
    b = np.zeros((n, 2 * h, 2 * w, m), np.float32)
    for p in range(2):
        for q in range(2):
            acc = np.zeros((n, h, w, m), np.float32)
            for dy in range(2):
                for dx in range(2):
                    acc += ap[:, p + dy:p + dy + h, q + dx:q + dx + w] @ kb[3 - p - 2 * dy, 3 - q - 2 * dx].T
            b[:, p::2, q::2] = acc
    b = round16(np.maximum(b * sb + tb, 0))
    return np.maximum(b @ kc * sc + tc, 0) + skip


@pytest.mark.parametrize("cin", [64, 128])
@pytest.mark.parametrize("shape", [(1, 4, 32), (2, 9, 21), (1, 13, 70), (3, 5, 33)])
def test_linknet_block_against_numpy(env, cin, shape):
    L, lib, dev = env
    n, h, w = shape
    m = cin // 4
    rng = np.random.default_rng(cin + h * 100 + w)
    x = round16(rng.normal(0, 1, (n, h, w, cin)))
    ka = round16(rng.normal(0, 0.15, (cin, m)))
    kb = round16(rng.normal(0, 0.15, (4, 4, m, m)))            # (kh, kw, cout, cin) as Keras stores it
    kc = round16(rng.normal(0, 0.25, (m, 64)))
    sa, sb_, sc = (rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (m, m, 64))
    ta, tb, tc = (rng.normal(0, 0.3, c).astype(np.float32) for c in (m, m, 64))
    skip = round16(rng.normal(0, 1, (n, 2 * h, 2 * w, 64)))
    want = reference(x, ka, kb, kc, sa, ta, sb_, tb, sc, tc, skip)

    # channel slices of wider buffers: input at channel 8 of cin + 24, skip at 16 of 96, output at 32 of 128
    xb = torch.zeros((n, h, w, cin + 24), dtype=h16(), device=dev)
    xb[..., 8:8 + cin] = dev_bf16(x, dev)
    sk = torch.zeros((n, 2 * h, 2 * w, 96), dtype=h16(), device=dev)
    sk[..., 16:80] = dev_bf16(skip, dev)
    ob = torch.full((n, 2 * h, 2 * w, 128), 7.0, dtype=h16(), device=dev)
    wa = dev_bf16(np.ascontiguousarray(ka.T), dev)                                   # [m][cin]
    wb = np.zeros((2, 2, m, 2, 2, m), np.float32)                                    # [a][b][cout][dy][dx][cin]
    for p in range(2):
        for q in range(2):
            for dy in range(2):
                for dx in range(2):
                    wb[p, q, :, dy, dx, :] = kb[3 - p - 2 * dy, 3 - q - 2 * dx]
    wbd = dev_bf16(wb, dev)
    wc = np.zeros((64, 32), np.float32)
    wc[:, :m] = kc.T
    wcd = dev_bf16(wc, dev)
    coef = [dev_f32(v, dev) for v in (sa, ta, sb_, tb, sc, tc)]

    d = L.LinknetBlockDesc()
    d.inp, d.wa, d.wb, d.wc = xb.data_ptr() + 2 * 8, wa.data_ptr(), wbd.data_ptr(), wcd.data_ptr()
    d.sa, d.ta, d.sb, d.tb, d.sc, d.tc = (c.data_ptr() for c in coef)
    d.skip, d.out = sk.data_ptr() + 2 * 16, ob.data_ptr() + 2 * 32
    d.N, d.H, d.W = n, h, w
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin + 24, (cin + 24) * w, (cin + 24) * w * h
    d.skip_pix_stride, d.skip_row_stride, d.skip_img_stride = 96, 96 * 2 * w, 96 * 4 * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 128, 128 * 2 * w, 128 * 4 * w * h
    d.fmt = L.FMT_F16 if h16() == torch.float16 else L.FMT_BF16
    d.cin = cin
    L.check(lib.miseg_linknet_block_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = ob[..., 32:96].float().cpu().numpy()
    tol = 2e-2 if h16() == torch.bfloat16 else 4e-3
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
    assert torch.all(ob[..., :32].float() == 7.0) and torch.all(ob[..., 96:].float() == 7.0), "wrote outside its slice"


def test_linknet_block_rejects_other_shapes(env):
    L, lib, dev = env
    t = torch.zeros(4096, dtype=torch.float32, device=dev)
    d = L.LinknetBlockDesc()
    for f in ("inp", "wa", "wb", "wc", "sa", "ta", "sb", "tb", "sc", "tc", "skip", "out"):
        setattr(d, f, t.data_ptr())
    d.N, d.H, d.W = 1, 4, 4
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 64, 256, 1024
    d.skip_pix_stride, d.skip_row_stride, d.skip_img_stride = 64, 512, 4096
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 64, 512, 4096
    d.cin = 96
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 96, 384, 1536
    assert lib.miseg_linknet_block_fwd(C.byref(d), None) == L.E_UNSUPPORTED_SHAPE
    d.cin = 64
    d.out_pix_stride = 60
    assert lib.miseg_linknet_block_fwd(C.byref(d), None) == L.E_INVALID_ARG
