"""Parity of the training-pass HIP kernels (through the C ABI) against torch-CPU autograd
and closed-form numpy on seeded inputs rounded to bf16.  Tolerances: fp32-accumulated
reductions 1e-4 relative; tensors stored in bf16 one bf16 ulp (2^-8 relative)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


@pytest.fixture(scope="module")
def env():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    L.check(lib.miseg_device_check())
    return L, lib, torch.device("cuda:0")


def dbf(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(dev)


def df32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


WG_CASES = [
    # n, h, w, cin, cout, k, stride, pad
    (2, 9, 13, 64, 64, 3, 1, 1),
    (1, 16, 24, 128, 256, 3, 1, 1),
    (1, 16, 24, 64, 128, 3, 2, 1),
    (2, 10, 14, 32, 64, 1, 2, 0),
    (1, 12, 12, 160, 96, 1, 1, 0),
    (1, 20, 12, 32, 8, 3, 1, 1),
    (3, 8, 8, 256, 128, 4, 2, 1),     # the strided 4x4 conv that is the data gradient of a transpose-conv
    (1, 70, 70, 64, 32, 3, 1, 1),     # several K-steps per split, rows wrapping inside a K-step
    (2, 17, 45, 32, 32, 3, 1, 1),     # partial pixel tiles in both directions
    (1, 24, 64, 32, 64, 3, 1, 1),
    (1, 33, 40, 128, 64, 3, 1, 1),    # two input-channel tiles
    (1, 16, 32, 96, 160, 3, 1, 1),    # ragged channel tiles
    (2, 9, 70, 64, 64, 1, 1, 0),
    (1, 8, 32, 32, 32, 2, 1, 1),
    (1, 11, 37, 64, 32, 2, 1, 0),
]


def _halo_ok(case):
    n, h, w, cin, cout, k, stride, pad = case
    return stride == 1 and k <= 3 and cin % 32 == 0 and pad < k


@pytest.mark.parametrize("hint", [0, 1, 2])
@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad(env, case, hint):
    L, lib, dev = env
    n, h, w, cin, cout, k, stride, pad = case
    if hint == 2 and not _halo_ok(case):
        pytest.skip("shape outside the halo kernel")
    rng = np.random.default_rng(hash(case) % 2**32)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    dy = bf16_round(rng.normal(0, 1, (n, ho, wo, cout)))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.zeros((cout, cin, k, k), requires_grad=True)
    out = F.conv2d(xt, wt, stride=stride, padding=pad)
    out.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    want = wt.grad.permute(0, 2, 3, 1).numpy()          # [Cout][KH][KW][Cin]
    xd, dyd = dbf(x, dev), dbf(dy, dev)
    dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=dev)
    d = L.WgradDesc()
    d.x, d.dy, d.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
    d.KH = d.KW = k
    d.stride_y = d.stride_x = stride
    d.pad_y = d.pad_x = pad
    d.x_pix, d.x_row, d.x_img = cin, cin * w, cin * w * h
    d.dy_pix, d.dy_row, d.dy_img = cout, cout * wo, cout * wo * ho
    d.tile_hint = hint
    L.check(lib.miseg_conv2d_wgrad(C.byref(d), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("hint", [0, 1])
@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout, kh, kw, sy, sx, py, px
    (2, 38, 24, 32, 64, 7, 1, 2, 1, 3, 0),     # the stem's form: 7x1 taps, stride 2 down the rows
    (1, 70, 40, 32, 64, 7, 1, 2, 1, 0, 0),     # several K-steps per block
    (1, 16, 20, 32, 32, 2, 2, 2, 2, 0, 0),     # four taps
    (3, 9, 11, 16, 24, 3, 1, 1, 1, 1, 0),      # ragged channels, three taps, a tap range with no wave work
    (1, 12, 16, 32, 64, 2, 4, 1, 2, 1, 1),     # eight taps
])
def test_wgrad_thin_layers_all_taps(env, case, hint):
    """Cin <= 32, Cout <= 64, 2..8 taps: hint 0 takes the kernel that stages dY once for all taps, hint 1 the
    per-tap kernel; both against torch autograd."""
    L, lib, dev = env
    n, h, w, cin, cout, kh, kw, sy, sx, py, px = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    ho = (h + 2 * py - kh) // sy + 1
    wo = (w + 2 * px - kw) // sx + 1
    dy = bf16_round(rng.normal(0, 1, (n, ho, wo, cout)))
    wt = torch.zeros((cout, cin, kh, kw), requires_grad=True)
    out = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), wt, stride=(sy, sx), padding=(py, px))
    out.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    want = wt.grad.permute(0, 2, 3, 1).numpy()          # [Cout][KH][KW][Cin]
    xd, dyd = dbf(x, dev), dbf(dy, dev)
    dw = torch.zeros((cout, kh, kw, cin), dtype=torch.float32, device=dev)
    d = L.WgradDesc()
    d.x, d.dy, d.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
    d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = kh, kw, sy, sx, py, px
    d.x_pix, d.x_row, d.x_img = cin, cin * w, cin * w * h
    d.dy_pix, d.dy_row, d.dy_img = cout, cout * wo, cout * wo * ho
    d.tile_hint = hint
    L.check(lib.miseg_conv2d_wgrad(C.byref(d), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("hint", [1, 2])
def test_wgrad_transpose_conv_phase_views(env, hint):
    """Weight gradient of Conv2DTranspose(4,2,'same'): one call per phase with dy viewed on the
    phase's lattice of the output gradient; result in the forward phase layout."""
    L, lib, dev = env
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 2, 6, 7, 64, 32
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    dout = bf16_round(rng.normal(0, 1, (n, 2 * h, 2 * w, cout)))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.zeros((cin, cout, 4, 4), requires_grad=True)
    F.conv_transpose2d(xt, wt, stride=2, padding=1).backward(torch.from_numpy(dout).permute(0, 3, 1, 2))
    gk = wt.grad.permute(2, 3, 1, 0).numpy()             # Keras (kh, kw, Cout, Cin)
    xd, dd = dbf(x, dev), dbf(dout, dev)
    dw = torch.zeros((2, 2, cout, 2, 2, cin), dtype=torch.float32, device=dev)
    for a in range(2):
        for b in range(2):
            d = L.WgradDesc()
            d.x = xd.data_ptr()
            d.dy = dd.data_ptr() + 2 * (a * 2 * w * cout + b * cout)
            d.dw = dw[a, b].data_ptr()
            d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
            d.KH = d.KW = 2
            d.stride_y = d.stride_x = 1
            d.pad_y, d.pad_x = 1 - a, 1 - b
            d.x_pix, d.x_row, d.x_img = cin, cin * w, cin * w * h
            d.dy_pix, d.dy_row, d.dy_img = 2 * cout, 2 * (2 * w * cout), 4 * h * w * cout
            d.tile_hint = hint
            L.check(lib.miseg_conv2d_wgrad(C.byref(d), None))
    torch.cuda.synchronize()
    got = dw.cpu().numpy()
    for a in range(2):
        for b in range(2):
            for dy_ in range(2):
                for dx_ in range(2):
                    np.testing.assert_allclose(got[a, b, :, dy_, dx_, :], gk[3 - a - 2 * dy_, 3 - b - 2 * dx_],
                                               rtol=1e-4, atol=1e-3)


def test_bn_stats_and_backward(env):
    L, lib, dev = env
    rng = np.random.default_rng(1)
    n, h, w, c, extra = 2, 11, 13, 40, 8
    pixels = n * h * w
    x = bf16_round(rng.normal(0.3, 1.5, (n, h, w, c)))
    gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
    beta = rng.normal(0, 0.5, c).astype(np.float32)
    mm0 = rng.normal(0, 1, c).astype(np.float32)
    mv0 = rng.uniform(0.5, 2, c).astype(np.float32)
    eps, mom = 1e-3, 0.99
    xbuf = torch.zeros((n, h, w, c + extra), dtype=torch.bfloat16, device=dev)
    xbuf[..., extra:] = dbf(x, dev)
    xptr = xbuf.data_ptr() + 2 * extra
    g, b_, mm, mv = df32(gamma, dev), df32(beta, dev), df32(mm0, dev), df32(mv0, dev)
    mean, var, scale, shift = (torch.zeros(c, device=dev) for _ in range(4))
    ws = torch.zeros(2 * c, dtype=torch.float64, device=dev)
    L.check(lib.miseg_bn_stats_fwd(xptr, pixels, c, c + extra, g.data_ptr(), b_.data_ptr(), mm.data_ptr(),
                                   mv.data_ptr(), mom, eps, mean.data_ptr(), var.data_ptr(),
                                   scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), None))
    x2 = x.reshape(-1, c).astype(np.float64)
    m_ref, v_ref = x2.mean(0), x2.var(0)
    np.testing.assert_allclose(mean.cpu().numpy(), m_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var.cpu().numpy(), v_ref, rtol=1e-5)
    sc_ref = gamma / np.sqrt(v_ref + eps)
    np.testing.assert_allclose(scale.cpu().numpy(), sc_ref, rtol=1e-5)
    np.testing.assert_allclose(shift.cpu().numpy(), beta - m_ref * sc_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(mm.cpu().numpy(), mom * mm0 + (1 - mom) * m_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv.cpu().numpy(), mom * mv0 + (1 - mom) * v_ref * pixels / (pixels - 1), rtol=1e-5)
    # backward of relu(bn(x)) against torch autograd (batch statistics)
    dy = bf16_round(rng.normal(0, 1, (n, h, w, c)))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    gt_, bt_ = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    y = F.relu(F.batch_norm(xt, None, None, gt_, bt_, True, 0.0, eps))
    y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    dyd = dbf(dy, dev)
    sums = torch.zeros(2 * c, dtype=torch.float64, device=dev)
    L.check(lib.miseg_bn_relu_bwd_reduce(xptr, dyd.data_ptr(), pixels, c, c + extra, c, scale.data_ptr(),
                                         shift.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, 1,
                                         sums.data_ptr(), None))
    dx = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_bn_relu_bwd_apply(xptr, dyd.data_ptr(), dx.data_ptr(), pixels, c, c + extra, c, c,
                                        scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                        eps, 1, sums.data_ptr(), None, 0, None))
    # the same with a gradient that is already there added in (fan-out of the layer's input)
    other = dbf(bf16_round(rng.normal(0, 1, (n, h, w, c))), dev)
    dx2 = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_bn_relu_bwd_apply(xptr, dyd.data_ptr(), dx2.data_ptr(), pixels, c, c + extra, c, c,
                                        scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                        eps, 1, sums.data_ptr(), other.data_ptr(), c, None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dx2.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy()
                               + other.float().cpu().numpy(), rtol=2e-2, atol=2e-2)
    s = sums.cpu().numpy()
    np.testing.assert_allclose(s[:c], bt_.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[c:], gt_.grad.numpy(), rtol=1e-4, atol=1e-3)
    want_dx = xt.grad.permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(dx.float().cpu().numpy(), want_dx, rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("shape", [(2, 16, 20), (1, 15, 21), (3, 97, 131), (1, 1, 3)])
def test_input_stats(env, shape):
    """bn_data's batch statistics of the fp32 input: groups of four pixels by 16-byte loads, the last
    pixels % 4 one by one."""
    L, lib, dev = env
    rng = np.random.default_rng(2)
    n, h, w = shape
    x = rng.uniform(0, 1, (n, h, w, 3)).astype(np.float32)
    beta = rng.normal(0, 0.1, 3).astype(np.float32)
    xd, b_ = df32(x, dev), df32(beta, dev)
    mm, mv = torch.zeros(3, device=dev), torch.ones(3, device=dev)
    mean, var, scale, shift = (torch.zeros(3, device=dev) for _ in range(4))
    ws = torch.zeros(6, dtype=torch.float64, device=dev)
    L.check(lib.miseg_input_stats_fwd(xd.data_ptr(), n * h * w, b_.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                      0.99, 2e-5, mean.data_ptr(), var.data_ptr(), scale.data_ptr(),
                                      shift.data_ptr(), ws.data_ptr(), None))
    x2 = x.reshape(-1, 3).astype(np.float64)
    np.testing.assert_allclose(mean.cpu().numpy(), x2.mean(0), rtol=1e-5)
    np.testing.assert_allclose(scale.cpu().numpy(), 1 / np.sqrt(x2.var(0) + 2e-5), rtol=1e-5)
    np.testing.assert_allclose(shift.cpu().numpy(), beta - x2.mean(0) / np.sqrt(x2.var(0) + 2e-5), rtol=1e-4, atol=1e-5)


def test_pool_and_upsample_backward(env):
    L, lib, dev = env
    rng = np.random.default_rng(3)
    n, h, w, c = 2, 12, 20, 16
    # max pool: many exact ties at zero, as after a ReLU
    x = bf16_round(np.maximum(rng.normal(0, 1, (n, h, w, c)), 0))
    dy = bf16_round(rng.normal(0, 1, (n, h // 2, w // 2, c)))
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    xp = F.pad(xt, (1, 1, 1, 1))
    F.max_pool2d(xp, 3, 2).backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    xd, dyd = dbf(x, dev), dbf(dy, dev)
    dx = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_maxpool3x3s2_bwd(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), n, h, w, c, c, c, c, None, 0, None))
    np.testing.assert_allclose(dx.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-2, atol=1e-2)
    held = bf16_round(rng.normal(0, 1, (n, h, w, c)))          # a gradient the input already holds
    hd = dbf(held, dev)
    L.check(lib.miseg_maxpool3x3s2_bwd(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), n, h, w, c, c, c, c,
                                       hd.data_ptr(), c, None))
    np.testing.assert_allclose(dx.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy() + held, rtol=1e-2,
                               atol=2e-2)
    # 2x2 max pool (VGG), ties at zero again; then the ReLU backward of a Conv2D(activation="relu")
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.max_pool2d(xt, 2, 2).backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    L.check(lib.miseg_maxpool2x2_bwd(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), n, h, w, c, c, c, c, None))
    np.testing.assert_array_equal(dx.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy())
    gfull = bf16_round(rng.normal(0, 1, (n, h, w, c)))
    addend = bf16_round(rng.normal(0, 1, (n, h, w, c)))
    gd, ad_ = dbf(gfull, dev), dbf(addend, dev)
    L.check(lib.miseg_relu_bwd(xd.data_ptr(), gd.data_ptr(), None, dx.data_ptr(), n * h * w, c, c, c, 0, c, None))
    np.testing.assert_array_equal(dx.float().cpu().numpy(), np.where(x > 0, gfull, 0))
    L.check(lib.miseg_relu_bwd(xd.data_ptr(), gd.data_ptr(), ad_.data_ptr(), dx.data_ptr(), n * h * w, c, c, c, c,
                               c, None))
    np.testing.assert_array_equal(dx.float().cpu().numpy(), bf16_round(np.where(x > 0, gfull, 0) + addend))
    # average pool
    L.check(lib.miseg_avgpool2x2_bwd(dyd.data_ptr(), dx.data_ptr(), n, h, w, c, c, c, None))
    want = np.repeat(np.repeat(dy, 2, 1), 2, 2) / 4
    np.testing.assert_array_equal(dx.float().cpu().numpy(), bf16_round(want))
    # nearest upsample x4
    g4 = bf16_round(rng.normal(0, 1, (n, 4 * h, 4 * w, c)))
    L.check(lib.miseg_upsample_nearest_bwd(dbf(g4, dev).data_ptr(), dx.data_ptr(), n, h, w, c, 2, c, c, None))
    want = g4.reshape(n, h, 4, w, 4, c).sum(axis=(2, 4))
    np.testing.assert_allclose(dx.float().cpu().numpy(), want, rtol=1e-2, atol=2e-2)
    # bilinear x2
    g2 = bf16_round(rng.normal(0, 1, (n, 2 * h, 2 * w, c)))
    xt = torch.zeros((n, c, h, w), requires_grad=True)
    F.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=False).backward(
        torch.from_numpy(g2).permute(0, 3, 1, 2))
    L.check(lib.miseg_upsample_bilinear2x_bwd(dbf(g2, dev).data_ptr(), dx.data_ptr(), n, h, w, c, c, c, None))
    np.testing.assert_allclose(dx.float().cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), rtol=1e-2, atol=2e-2)
    # channel-slice copy with accumulate, channel sums
    a = bf16_round(rng.normal(0, 1, (n, h, w, c)))
    b = bf16_round(rng.normal(0, 1, (n, h, w, 2 * c)))
    ad, bd = dbf(a, dev), dbf(b, dev)
    L.check(lib.miseg_copy_channels(ad.data_ptr(), bd.data_ptr() + 2 * c, bd.data_ptr() + 2 * c, n * h * w, c,
                                    c, 2 * c, 2 * c, None))
    got = bd.float().cpu().numpy()
    np.testing.assert_array_equal(got[..., :c], b[..., :c])
    np.testing.assert_array_equal(got[..., c:], bf16_round(a + b[..., c:]))
    sums = torch.zeros(c, dtype=torch.float64, device=dev)
    L.check(lib.miseg_channel_sums(ad.data_ptr(), n * h * w, c, c, sums.data_ptr(), None))
    np.testing.assert_allclose(sums.cpu().numpy(), a.reshape(-1, c).astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("classes,temp", [(3, 1.0), (3, 0.1), (2, 1.0), (4, 1.0)])
def test_loss_grad_matches_autograd(env, classes, temp):
    L, lib, dev = env
    rng = np.random.default_rng(4)
    n, h, w = 2, 10, 12
    z = rng.normal(0, 1.5, (n, h, w, classes)).astype(np.float32)
    gt = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, (n, h, w))]
    zt = torch.from_numpy(z).requires_grad_(True)
    pr = torch.softmax(zt / temp, dim=-1)
    g = torch.from_numpy(gt)
    q = torch.clamp(pr / pr.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
    cce = torch.mean(-g * torch.log(q))
    tp = (g * pr).sum((0, 1, 2))
    score = (2 * tp + 1e-5) / (g.sum((0, 1, 2)) + pr.sum((0, 1, 2)) + 1e-5)
    loss = 0.5 * (cce + 1 - score.mean())
    loss.backward()
    prd, gtd = df32(pr.detach().numpy(), dev), df32(gt, dev)
    sums = torch.zeros(4 * classes, dtype=torch.float64, device=dev)
    L.check(lib.miseg_loss_sums(gtd.data_ptr(), prd.data_ptr(), sums.data_ptr(), n * h * w, classes, None))
    want = zt.grad.numpy()
    for stride in (8, 32, 5):          # multiples of 8: padding channels are zeroed by the kernel
        dz = torch.full((n, h, w, stride), 7.0, dtype=torch.bfloat16, device=dev)
        if stride < classes:
            continue
        L.check(lib.miseg_loss_grad(gtd.data_ptr(), prd.data_ptr(), sums.data_ptr(), dz.data_ptr(), n * h * w,
                                    classes, stride, 0.5, temp, 1, 1, None))
        torch.cuda.synchronize()
        got = dz.float().cpu().numpy()
        np.testing.assert_allclose(got[..., :classes], want, rtol=2e-2, atol=2e-3 * np.abs(want).max())
        assert np.all(got[..., classes:] == (0 if stride % 8 == 0 else 7.0))


def test_adam_keras_form(env):
    L, lib, dev = env
    rng = np.random.default_rng(6)
    cnt = 10007
    p0 = rng.normal(0, 1, cnt).astype(np.float32)
    g = rng.normal(0, 0.1, cnt).astype(np.float32)
    m0 = rng.normal(0, 0.01, cnt).astype(np.float32)
    v0 = rng.uniform(0, 0.01, cnt).astype(np.float32)
    lr, b1, b2, eps, t = 1e-3, 0.9, 0.999, 1e-7, 3
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    p, m, v, gd = df32(p0, dev), df32(m0, dev), df32(v0, dev), df32(g, dev)
    pb = torch.zeros(cnt, dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_adam_step(p.data_ptr(), m.data_ptr(), v.data_ptr(), gd.data_ptr(), pb.data_ptr(), cnt,
                                lr_t, b1, b2, eps, 2.0, None))
    torch.cuda.synchronize()
    g2 = g * 2.0
    m1 = b1 * m0 + (1 - b1) * g2
    v1 = b2 * v0 + (1 - b2) * g2 * g2
    p1 = p0 - lr_t * m1 / (np.sqrt(v1) + eps)
    # the device contracts a*b+c into fma: a few fp32 ulps of difference
    np.testing.assert_allclose(m.cpu().numpy(), m1, rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(v.cpu().numpy(), v1, rtol=2e-5, atol=1e-10)
    np.testing.assert_allclose(p.cpu().numpy(), p1, rtol=2e-5, atol=1e-6)
    np.testing.assert_array_equal(pb.float().cpu().numpy(), bf16_round(p.cpu().numpy()))


@pytest.mark.parametrize("case", [(2, 16, 40, 64, 128, 3, False), (1, 9, 33, 64, 64, 3, False),
                                  (2, 12, 20, 32, 32, 3, False), (1, 8, 16, 64, 32, 1, False),
                                  (2, 8, 24, 64, 64, 4, True), (1, 10, 20, 128, 160, 3, False),
                                  (3, 250, 400, 32, 32, 3, False)])
def test_conv_epilogue_statistics(env, case):
    """miseg_conv_desc.stats: per-channel sum / sum of squares of the STORED bf16 output, spread over
    replicas, then miseg_bn_finalize == miseg_bn_stats_fwd on the same tensor (mean/var to fp32
    round-off; the fused path adds the same bf16 values in a different order)."""
    L, lib, dev = env
    n, h, w, cin, cout, k, transposed = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = torch.from_numpy(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)).to(torch.bfloat16).to(dev)
    R = 8
    d = L.ConvDesc()
    if transposed:
        wt = torch.from_numpy(rng.normal(0, 0.1, (4, cout, 2, 2, cin)).astype(np.float32)).to(torch.bfloat16).to(dev)
        oh, ow = 2 * h, 2 * w
        d.Ho, d.Wo, d.KH, d.KW, d.pad_y, d.pad_x, d.transposed = h, w, 2, 2, 0, 0, 1
    else:
        wt = torch.from_numpy(rng.normal(0, 0.1, (cout, k, k, cin)).astype(np.float32)).to(torch.bfloat16).to(dev)
        oh, ow = h, w
        d.Ho, d.Wo, d.KH, d.KW, d.pad_y, d.pad_x = h, w, k, k, (k - 1) // 2, (k - 1) // 2
    out = torch.zeros((n, oh, ow, cout), dtype=torch.bfloat16, device=dev)
    stats = torch.zeros((R + 1, 2, cout), dtype=torch.float64, device=dev)
    d.inp, d.weight, d.out = x.data_ptr(), wt.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Cout = n, h, w, cin, cout
    d.stride_y = d.stride_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * ow, cout * ow * oh
    d.temperature = 1.0
    d.stats, d.stats_replicas, d.stats_stride = stats.data_ptr(), R, cout
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    o = out.double().reshape(-1, cout)
    got = stats[:R].sum(0).cpu().numpy()
    np.testing.assert_allclose(got[0], o.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(got[1], (o * o).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # finalize == the stand-alone statistics kernel
    beta = torch.zeros(cout, device=dev)
    res = []
    for fused in (True, False):
        mean, var, scale, shift = (torch.zeros(cout, device=dev) for _ in range(4))
        mm, mv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        if fused:
            L.check(lib.miseg_bn_finalize(stats.data_ptr(), R, cout, o.shape[0], cout, None, beta.data_ptr(),
                                          mm.data_ptr(), mv.data_ptr(), 0.99, 1e-3, mean.data_ptr(),
                                          var.data_ptr(), scale.data_ptr(), shift.data_ptr(), None))
        else:
            ws = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
            L.check(lib.miseg_bn_stats_fwd(out.data_ptr(), o.shape[0], cout, cout, None, beta.data_ptr(),
                                           mm.data_ptr(), mv.data_ptr(), 0.99, 1e-3, mean.data_ptr(),
                                           var.data_ptr(), scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), None))
        torch.cuda.synchronize()
        res.append([t.cpu().numpy() for t in (mean, var, scale, shift, mm, mv)])
    for a, b in zip(*res):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("classes,flags,temp", [(3, 4, 1.0), (3, 6, 1.0), (3, 6, 0.1), (2, 4, 1.0), (1, 2, 1.0),
                                                (1, 6, 1.0), (1, 3, 1.0)])
def test_loss_grad_ex_focal_and_sigmoid_match_autograd(env, classes, flags, temp):
    """miseg_loss_grad_ex / miseg_loss_sums_focal: focal loss (segmentation_models categorical_focal_loss,
    alpha 0.25, gamma 2; utils/model.py:94-95) and the sigmoid head (classes == 1, utils/model.py:180)
    against torch autograd of the formulas in oracle/losses.py."""
    L, lib, dev = env
    rng = np.random.default_rng(40 + classes + flags)
    n, h, w = 2, 10, 12
    alpha, gamma = 0.25, 2.0
    z = rng.normal(0, 1.5, (n, h, w, classes)).astype(np.float32)
    if classes == 1:
        gt = rng.integers(0, 2, (n, h, w, 1)).astype(np.float32)
    else:
        gt = np.eye(classes, dtype=np.float32)[rng.integers(0, classes, (n, h, w))]
    zt = torch.from_numpy(z).requires_grad_(True)
    pr = torch.sigmoid(zt) if classes == 1 else torch.softmax(zt / temp, dim=-1)
    g = torch.from_numpy(gt)
    loss = 0.0
    if flags & 1:
        q = torch.clamp(pr / pr.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
        loss = loss + torch.mean(-g * torch.log(q))
    if flags & 2:
        tp = (g * pr).sum((0, 1, 2))
        loss = loss + 1 - ((2 * tp + 1e-5) / (g.sum((0, 1, 2)) + pr.sum((0, 1, 2)) + 1e-5)).mean()
    focal_val = None
    if flags & 4:
        q = torch.clamp(pr, 1e-7, 1 - 1e-7)
        fl = torch.mean(-g * alpha * (1 - q) ** gamma * torch.log(q))
        focal_val = float(fl.detach())
        loss = loss + fl
    (0.5 * loss).backward()
    want = zt.grad.numpy()
    prd, gtd = df32(pr.detach().numpy(), dev), df32(gt, dev)
    sums = torch.zeros(5 * classes, dtype=torch.float64, device=dev)
    L.check(lib.miseg_loss_sums_focal(gtd.data_ptr(), prd.data_ptr(), sums.data_ptr(), n * h * w, classes,
                                      alpha, gamma, None))
    torch.cuda.synchronize()
    s = sums.cpu().numpy().reshape(5, classes)
    if focal_val is not None:
        assert abs(-s[4].sum() / gt.size - focal_val) <= 1e-5 * max(1.0, abs(focal_val))
    ref4 = torch.zeros(4 * classes, dtype=torch.float64, device=dev)
    L.check(lib.miseg_loss_sums(gtd.data_ptr(), prd.data_ptr(), ref4.data_ptr(), n * h * w, classes, None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(s[:4].ravel(), ref4.cpu().numpy(), rtol=1e-12)
    dz = torch.full((n, h, w, 8), 7.0, dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_loss_grad_ex(gtd.data_ptr(), prd.data_ptr(), sums.data_ptr(), dz.data_ptr(), n * h * w,
                                   classes, 8, 0.5, temp, flags, alpha, gamma, None))
    torch.cuda.synchronize()
    got = dz.float().cpu().numpy()
    np.testing.assert_allclose(got[..., :classes], want, rtol=2e-2, atol=2e-3 * np.abs(want).max())
    assert np.all(got[..., classes:] == 0)


MASK_CASES = [
    # n, h, w, cin, cout, k, stride, pad, mask_c, kernel the launcher picks
    (2, 20, 40, 64, 128, 3, 1, 1, 128, "conv3x3_halo_pw_kernel"),     # ring kernel, every channel masked
    (1, 17, 45, 128, 256, 3, 1, 1, 128, "conv3x3_halo_pw_kernel"),    # ragged patches; a skip tensor behind the mask
    (2, 16, 32, 64, 64, 3, 1, 1, 64, "conv3x3_c64p_kernel<true, true>"),   # persistent 64-channel kernel
    (1, 19, 45, 64, 48, 3, 1, 1, 40, "conv3x3_c64p_kernel<true, true>"),   # ragged patches, cout group and mask
    (2, 16, 32, 128, 64, 3, 1, 1, 64, "conv3x3_halo_ks_kernel"),       # K-split kernel
    (2, 19, 70, 32, 32, 3, 1, 1, 32, "conv3x3_c32p_kernel"),          # persistent kernel: x loads issued by hand
    (1, 16, 64, 32, 32, 3, 1, 1, 16, "conv3x3_c32p_kernel"),          # the second octet of a pixel unmasked
    (2, 24, 24, 32, 64, 4, 2, 1, 64, "conv_igemm_kernel"),            # the generic tiles (a transpose-conv's gradient)
    (1, 16, 16, 64, 64, 1, 1, 0, 64, "conv_igemm_dma_kernel"),        # 1x1 on the LDS-DMA tiles
]


@pytest.mark.parametrize("case", MASK_CASES)
def test_conv_bn_relu_mask_mode(env, case):
    """miseg_conv_desc.res_mode = MISEG_RES_BN_RELU_MASK: a data gradient that also applies the mask of the
    ReLU behind it and gathers the BatchNormalization's two backward sums (the tf.keras FusedBatchNormGrad
    reductions of the model.fit step, /root/reference model_train.py:284-329), against numpy on the same
    bf16-rounded operands; miseg_bn_bwd_sums finishes the sums."""
    L, lib, dev = env
    n, h, w, cin, cout, k, stride, pad, mc, kernel = case
    rng = np.random.default_rng(hash(case[:9]) % 2**32)
    dy = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (cout, k, k, cin)))
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = bf16_round(rng.normal(0.3, 1.5, (n, ho, wo, mc)))           # the BatchNorm's input at the output positions
    scale = rng.uniform(0.5, 1.5, mc).astype(np.float32) * rng.choice([-1.0, 1.0], mc).astype(np.float32)
    shift = rng.normal(0, 0.5, mc).astype(np.float32)
    acc = F.conv2d(torch.from_numpy(dy).permute(0, 3, 1, 2), torch.from_numpy(wt).permute(0, 3, 1, 2), stride=stride,
                   padding=pad).permute(0, 2, 3, 1).numpy()
    want = acc.copy()
    want[..., :mc] *= (x * scale + shift > 0)
    want = bf16_round(want)
    dyd, wd, xd = dbf(dy, dev), dbf(wt, dev), dbf(x, dev)
    out = torch.full((n, ho, wo, cout), 7.0, dtype=torch.bfloat16, device=dev)
    sc_d, sh_d = df32(scale, dev), df32(shift, dev)
    reps, stride_c = 32, (mc + 31) // 32 * 32
    stats = torch.zeros((reps + 1, 2, stride_c), dtype=torch.float64, device=dev)
    d = L.ConvDesc()
    d.inp, d.weight, d.out = dyd.data_ptr(), wd.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
    d.KH = d.KW = k
    d.stride_y = d.stride_x = stride
    d.pad_y = d.pad_x = pad
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * wo, cout * wo * ho
    d.out_dtype, d.res_dtype, d.act, d.temperature = L.DT_BF16, L.DT_BF16, L.ACT_NONE, 1.0
    d.residual = xd.data_ptr()
    d.res_pix_stride, d.res_row_stride, d.res_img_stride = mc, mc * wo, mc * wo * ho
    d.scale, d.shift = sc_d.data_ptr(), sh_d.data_ptr()
    d.res_mode, d.res_mask_c = L.RES_BN_RELU_MASK, mc
    d.stats, d.stats_replicas, d.stats_stride = stats.data_ptr(), reps, stride_c
    buf = C.create_string_buffer(128)
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode().startswith(kernel), buf.value
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    tol = 2.0 ** -7 * np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
    flips = np.abs(got - want) > tol
    # a value within rounding of zero on the mask's boundary cannot flip: the mask comes from x, not from the result
    assert flips.mean() < 1e-4, (flips.mean(), np.abs(got - want).max())
    assert np.array_equal(got[..., :mc] == 0, (want[..., :mc] == 0) | (got[..., :mc] == 0))
    assert np.all(got[..., :mc][(x * scale + shift) <= 0] == 0)
    # the two sums, of the values as stored
    s1 = got[..., :mc].astype(np.float64).sum(axis=(0, 1, 2))
    s2 = (got[..., :mc].astype(np.float64) * x).sum(axis=(0, 1, 2))
    tot = stats[:reps].sum(dim=0).cpu().numpy()
    np.testing.assert_allclose(tot[0, :mc], s1, rtol=1e-5, atol=1e-4 * np.abs(s1).max())
    np.testing.assert_allclose(tot[1, :mc], s2, rtol=1e-5, atol=1e-4 * np.abs(s2).max())
    mean, var, eps = x.mean(axis=(0, 1, 2)).astype(np.float32), x.var(axis=(0, 1, 2)).astype(np.float32), 1e-3
    sums = torch.empty(2 * mc, dtype=torch.float64, device=dev)
    mean_d, var_d = df32(mean, dev), df32(var, dev)
    L.check(lib.miseg_bn_bwd_sums(stats.data_ptr(), reps, stride_c, mc, mean_d.data_ptr(), var_d.data_ptr(), eps,
                                  sums.data_ptr(), None))
    torch.cuda.synchronize()
    sm = sums.cpu().numpy()
    np.testing.assert_allclose(sm[:mc], s1, rtol=1e-5, atol=1e-4 * np.abs(s1).max())
    ref2 = (s2 - mean.astype(np.float64) * s1) / np.sqrt(var.astype(np.float64) + eps)
    np.testing.assert_allclose(sm[mc:], ref2, rtol=1e-4, atol=1e-4 * np.abs(ref2).max())


def test_conv_bn_relu_mask_mode_is_declined_where_it_is_not_built(env):
    """fp16 storage (inference only) has no masked kernels: the call reports MISEG_E_UNSUPPORTED_SHAPE instead of
    running without the mask."""
    L, lib, dev = env
    t = torch.zeros(1 << 16, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(256, dtype=torch.float32, device=dev)
    st = torch.zeros(33 * 2 * 64, dtype=torch.float64, device=dev)
    d = L.ConvDesc()
    d.inp = d.weight = d.out = d.residual = t.data_ptr()
    d.scale = d.shift = f.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = 1, 8, 8, 64, 8, 8, 64
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 64, 512, 4096
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 64, 512, 4096
    d.res_pix_stride, d.res_row_stride, d.res_img_stride = 64, 512, 4096
    d.out_dtype, d.res_dtype, d.act, d.temperature = L.DT_BF16, L.DT_BF16, L.ACT_NONE, 1.0
    d.res_mode, d.res_mask_c = L.RES_BN_RELU_MASK, 64
    d.stats, d.stats_replicas, d.stats_stride = st.data_ptr(), 32, 64
    d.fmt = L.FMT_F16
    assert lib.miseg_conv2d_fwd(C.byref(d), None) == L.E_UNSUPPORTED_SHAPE
    d.fmt = L.FMT_BF16
    d.res_mask_c = 12                      # not a multiple of 8
    assert lib.miseg_conv2d_fwd(C.byref(d), None) == L.E_INVALID_ARG


@pytest.mark.parametrize("case", [(2, 19, 70, 32, 0), (1, 8, 32, 32, 0), (1, 37, 45, 16, 0), (2, 16, 64, 3, 1),
                                  (1, 21, 33, 3, 1)])
def test_conv_c32p_prologue(env, case):
    """miseg_conv_desc.in_scale / in_shift on a 32-channel 3x3 layer: the persistent kernel applies
    relu(x * scale + shift) -- the BatchNormalization + ReLU pair in front of the convolution
    (/root/reference utils/model.py:176-186 builds Conv-BN-ReLU chains) -- to the halo tile in LDS, so that
    the activated tensor is never written.  Against the two steps done one after the other on the same
    bf16-rounded operands; the zero padding stays zero (it is not bn(0)); with statistics of the stored output and
    as the softmax head."""
    L, lib, dev = env
    n, h, w, cout, head = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = bf16_round(rng.normal(0.2, 1.0, (n, h, w, 32)))
    wt = bf16_round(rng.normal(0, 0.1, (cout, 3, 3, 32)))
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32) * rng.choice([-1.0, 1.0], 32).astype(np.float32)
    shift = rng.normal(0.3, 0.5, 32).astype(np.float32)              # bn(0) != 0: the padding must not see it
    a = bf16_round(np.maximum(x * scale + shift, 0))
    acc = F.conv2d(torch.from_numpy(a).permute(0, 3, 1, 2), torch.from_numpy(wt).permute(0, 3, 1, 2),
                   padding=1).permute(0, 2, 3, 1).numpy()
    xd, wd = dbf(x, dev), dbf(wt, dev)
    sc_d, sh_d = df32(scale, dev), df32(shift, dev)
    d = L.ConvDesc()
    d.inp, d.weight = xd.data_ptr(), wd.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, 32, h, w, cout
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 32, 32 * w, 32 * w * h
    d.in_scale, d.in_shift = sc_d.data_ptr(), sh_d.data_ptr()
    d.temperature = 1.0
    if head:
        bias = rng.normal(0, 0.3, cout).astype(np.float32)
        bias_d = df32(bias, dev)
        out = torch.full((n, h, w, cout), 7.0, dtype=torch.float32, device=dev)
        d.out, d.out_dtype, d.act, d.shift = out.data_ptr(), L.DT_F32, L.ACT_SOFTMAX, bias_d.data_ptr()
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * w, cout * w * h
        z = acc + bias
        e = np.exp(z - z.max(axis=-1, keepdims=True))
        want = e / e.sum(axis=-1, keepdims=True)
    else:
        out = torch.full((n, h, w, cout), 7.0, dtype=torch.bfloat16, device=dev)
        reps, stride_c = 32, 32
        stats = torch.zeros((reps + 1, 2, stride_c), dtype=torch.float64, device=dev)
        d.out, d.out_dtype, d.act = out.data_ptr(), L.DT_BF16, L.ACT_NONE
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * w, cout * w * h
        d.stats, d.stats_replicas, d.stats_stride = stats.data_ptr(), reps, stride_c
        want = bf16_round(acc)
    buf = C.create_string_buffer(128)
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode() == "conv3x3_c32p_kernel<false, true>", buf.value
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    if head:
        assert np.abs(got - want).max() < 2e-3
    else:
        tol = 2.0 ** -7 * np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
        assert (np.abs(got - want) > tol).mean() < 1e-4, np.abs(got - want).max()
        tot = stats[:reps].sum(dim=0).cpu().numpy()
        s1 = got.astype(np.float64).sum(axis=(0, 1, 2))
        s2 = (got.astype(np.float64) ** 2).sum(axis=(0, 1, 2))
        np.testing.assert_allclose(tot[0, :cout], s1, rtol=1e-5, atol=1e-4 * np.abs(s1).max())
        np.testing.assert_allclose(tot[1, :cout], s2, rtol=1e-5, atol=1e-4 * np.abs(s2).max())
    # fp16 storage has no such instantiation: the layer goes to the generic prologue kernel there
    d.fmt = L.FMT_F16
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert "c32p" not in buf.value.decode(), buf.value


@pytest.mark.parametrize("case", [(2, 17, 45, 32), (1, 8, 32, 8), (1, 40, 70, 16)])
def test_wgrad_x_prologue(env, case):
    """miseg_wgrad_desc.x_scale / x_shift: the weight gradient of a 3x3 layer whose input relu(bn(x)) was never
    stored (its forward pass ran with miseg_conv_desc.in_scale): the halo kernel activates the X tile in LDS.
    Against autograd on the activated, bf16-rounded tensor; pixels outside the image are padding, not bn(0).
    Shapes the prologue is not built for are declined."""
    L, lib, dev = env
    n, h, w, cout = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = bf16_round(rng.normal(0.2, 1.0, (n, h, w, 32)))
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32) * rng.choice([-1.0, 1.0], 32).astype(np.float32)
    shift = rng.normal(0.3, 0.5, 32).astype(np.float32)
    a = bf16_round(np.maximum(x * scale + shift, 0))
    dy = bf16_round(rng.normal(0, 1, (n, h, w, cout)))
    wt = torch.zeros((cout, 32, 3, 3), requires_grad=True)
    F.conv2d(torch.from_numpy(a).permute(0, 3, 1, 2), wt, padding=1).backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    want = wt.grad.permute(0, 2, 3, 1).numpy()
    xd, dyd = dbf(x, dev), dbf(dy, dev)
    sc_d, sh_d = df32(scale, dev), df32(shift, dev)
    dw = torch.zeros((cout, 3, 3, 32), dtype=torch.float32, device=dev)
    d = L.WgradDesc()
    d.x, d.dy, d.dw = xd.data_ptr(), dyd.data_ptr(), dw.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, 32, h, w, cout
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.x_pix, d.x_row, d.x_img = 32, 32 * w, 32 * w * h
    d.dy_pix, d.dy_row, d.dy_img = cout, cout * w, cout * w * h
    d.x_scale, d.x_shift = sc_d.data_ptr(), sh_d.data_ptr()
    L.check(lib.miseg_conv2d_wgrad(C.byref(d), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
    d.tile_hint = 1                                        # the per-tap kernel has no prologue
    assert lib.miseg_conv2d_wgrad(C.byref(d), None) == L.E_UNSUPPORTED_SHAPE
    d.tile_hint, d.x_shift = 0, None
    assert lib.miseg_conv2d_wgrad(C.byref(d), None) == L.E_INVALID_ARG
