"""Parity of each HIP kernel, called through the C ABI, against the CPU oracle
(oracle/ops_np.py, oracle/ops_torch.py) on seeded inputs.  Inputs and weights
are rounded to the 16-bit storage format first, so the oracle sees exactly the
operands the kernels see and the remaining difference is fp32 accumulation order
plus the rounding of the stored result (tolerances stated per test, sized for
bf16, the coarser of the two formats).

Every test of this module runs twice: on bf16 storage (MISEG_FMT_BF16, the format of
the training path) and on fp16 storage (MISEG_FMT_F16, the inference default); the
helper names keep their historical "bf16" but follow the format under test."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# the storage format under test (switched by the autouse fixture below)
STORAGE = {"dtype": torch.bfloat16, "fmt": 0, "suffix": ""}


def h16():
    return STORAGE["dtype"]


@pytest.fixture(autouse=True, params=["bf16", "fp16"])
def storage(request):
    if request.param == "fp16":
        STORAGE.update(dtype=torch.float16, fmt=1, suffix="_f16")
    else:
        STORAGE.update(dtype=torch.bfloat16, fmt=0, suffix="")
    yield request.param
    STORAGE.update(dtype=torch.bfloat16, fmt=0, suffix="")


def new_desc(L):
    d = L.ConvDesc()
    d.fmt = STORAGE["fmt"]
    return d


def fn(lib, name):
    """Entry point `name` for the format under test (the layers between convolutions have _f16 twins)."""
    return getattr(lib, name + STORAGE["suffix"])


def flips(rate):
    """Allowed share of stored values that differ by one ulp between two correct code paths whose
    fp32 results differ in the last bits (summation order, fused against separate multiply-add):
    `rate` is the bf16 figure; fp16 has 8x finer ulps, so an fp32 difference crosses a rounding
    boundary 8x as often."""
    return rate * (8 if STORAGE["fmt"] == 1 else 1)


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(h16()).float().numpy()


@pytest.fixture(scope="module")
def env():
    import miseg_amd
    from miseg_amd import lib as L
    lib = L.load()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    L.check(lib.miseg_device_check())
    return L, lib, torch.device("cuda:0")


def dev_bf16(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(h16()).to(dev)


def dev_f32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


def run_conv(env, x, w_hwio, stride=1, pad=0, scale=None, shift=None, act=0, residual=None,
             res_f32=True, scale2=None, shift2=None, in_scale=None, in_shift=None, out_f32=False,
             want_out=True, tile_hint=0, in_extra_c=0, out_extra_c=0, res_after=False, res_shift=0, out2_linear=False):
    """x: (N,H,W,Cin) fp32 (bf16-representable); returns (out, out2) as fp32 numpy."""
    L, lib, dev = env
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w_hwio.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    xin = torch.zeros((n, h, wd, cin + in_extra_c), dtype=h16(), device=dev)
    xin[..., in_extra_c:] = dev_bf16(x, dev)
    wt = dev_bf16(w_hwio.transpose(3, 0, 1, 2), dev)
    out = torch.full((n, ho, wo, cout + out_extra_c), 7.0,
                     dtype=torch.float32 if out_f32 else h16(), device=dev)
    out2 = torch.full((n, ho, wo, cout), 7.0, dtype=h16(), device=dev)
    keep = []

    def f32p(a):
        if a is None:
            return None
        t = dev_f32(a, dev)
        keep.append(t)
        return t.data_ptr()

    d = new_desc(L)
    d.inp = xin.data_ptr() + 2 * in_extra_c
    d.weight = wt.data_ptr()
    d.N, d.H, d.W, d.Cin = n, h, wd, cin
    d.Ho, d.Wo, d.Cout = ho, wo, cout
    d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = kh, kw, stride, stride, pad, pad
    cs = cin + in_extra_c
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cs, cs * wd, cs * wd * h
    if want_out:
        d.out = out.data_ptr() + out_extra_c * out.element_size()
        co = cout + out_extra_c
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = co, co * wo, co * wo * ho
    d.out_dtype = L.DT_F32 if out_f32 else L.DT_BF16
    if scale2 is not None:
        d.out2 = out2.data_ptr()
        d.out2_pix_stride, d.out2_row_stride, d.out2_img_stride = cout, cout * wo, cout * wo * ho
        d.scale2, d.shift2 = f32p(scale2), f32p(shift2)
    if residual is not None:
        r = dev_f32(residual, dev) if res_f32 else dev_bf16(residual, dev)
        keep.append(r)
        d.residual = r.data_ptr()
        d.res_dtype = L.DT_F32 if res_f32 else L.DT_BF16
        rh, rw = r.shape[1], r.shape[2]
        d.res_pix_stride, d.res_row_stride, d.res_img_stride = cout, cout * rw, cout * rw * rh
        d.res_upsample_shift = res_shift
    d.scale, d.shift = f32p(scale), f32p(shift)
    d.in_scale, d.in_shift = f32p(in_scale), f32p(in_shift)
    d.act = act
    d.temperature = 1.0
    d.tile_hint = tile_hint
    d.res_after_act = 1 if res_after else 0
    d.out2_linear = 1 if out2_linear else 0
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    o = out[..., out_extra_c:].float().cpu().numpy() if want_out else None
    o2 = out2.float().cpu().numpy() if scale2 is not None else None
    if out_extra_c and want_out:
        assert torch.all(out[..., :out_extra_c].float() == 7.0), "wrote outside its channel slice"
    return o, o2


def ref_conv(x, w, stride, pad, in_scale=None, in_shift=None):
    from oracle import ops_torch as T
    xt = T.asarray(x)
    if in_scale is not None:
        xt = torch.relu(xt * torch.from_numpy(in_scale) + torch.from_numpy(in_shift))
        xt = xt.to(h16()).float()     # the kernel re-rounds the activated operand
    return T.to_numpy(T.conv2d(xt, w, stride=stride, pad=pad))


CASES = [
    # n, h, w, cin, cout, k, stride, pad
    (1, 12, 20, 32, 32, 3, 1, 1),
    (2, 9, 13, 64, 64, 3, 1, 1),       # odd sizes, M tail
    (1, 16, 24, 64, 128, 3, 2, 1),     # stride 2, symmetric pad
    (1, 16, 16, 96, 160, 1, 1, 0),     # cout not a tile multiple
    (1, 10, 14, 32, 64, 1, 2, 0),      # shortcut conv
    (1, 8, 8, 128, 256, 3, 1, 1),
    (1, 20, 12, 32, 8, 3, 1, 1),       # narrow cout
    (1, 20, 70, 64, 96, 3, 1, 1),      # several patches per row, ragged right/bottom edge
    (2, 17, 33, 128, 128, 3, 1, 1),    # two channel chunks, odd sizes
    (1, 8, 32, 192, 64, 3, 1, 1),      # exactly one patch, three chunks
    (2, 50, 70, 128, 64, 3, 1, 1),     # Cout <= 64: waves split K; two chunks, ragged patches
    (1, 33, 40, 192, 48, 3, 1, 1),     # the same with a ragged cout group
    (2, 11, 37, 64, 40, 3, 1, 1),      # ragged octet inside the second cout group
    (1, 19, 33, 128, 36, 3, 1, 1),     # Cout not a multiple of 8: generic epilogue after the accumulator swap
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("hint", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 32])
def test_conv_plain(env, case, hint):
    """hint & 15 picks the tile shape (8..10: halo-resident 3x3 variants, used where the
    layer qualifies); hint & 16 forces the register-staged path where an LDS-DMA path
    (Cin % 64 == 0) would be chosen; hint & 32 keeps 3x3 layers off the halo kernel."""
    n, h, w, cin, cout, k, stride, pad = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (k, k, cin, cout)))
    got, _ = run_conv(env, x, wt, stride, pad, out_f32=True, tile_hint=hint)
    want = ref_conv(x, wt, stride, pad)
    # fp32 accumulation of identical bf16 operands: only summation order differs
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("shape", [(1, 1, 64, 40), (1, 1, 96, 128), (3, 3, 64, 64), (3, 3, 128, 128), (3, 3, 32, 16)])
@pytest.mark.parametrize("res_f32", [False, True])
def test_conv_residual_after_activation(env, shape, res_f32):
    """out = relu(conv*scale + shift) + residual: the Add([relu(bn(conv)), skip]) of the LinkNet decoder
    blocks (segmentation_models DecoderBlock, /root/reference utils/model.py:188-198), on the generic
    and on the direct epilogue."""
    k, _, cin, cout = shape
    rng = np.random.default_rng(cin * 1000 + cout + k)
    n, h, w = 2, 11, 37
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (k, k, cin, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    res = rng.normal(0, 1, (n, h, w, cout)).astype(np.float32)
    if not res_f32:
        res = bf16_round(res)
    base = ref_conv(x, wt, 1, k // 2)
    want = np.maximum(base * sc + sh, 0) + res
    for out_f32 in (False, True):
        got, _ = run_conv(env, x, wt, 1, k // 2, scale=sc, shift=sh, act=1, residual=res, res_f32=res_f32,
                          out_f32=out_f32, res_after=True)
        tol = 1e-4 if out_f32 else 1e-2
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
    # the pre-activation form differs wherever the convolution's value is negative
    pre, _ = run_conv(env, x, wt, 1, k // 2, scale=sc, shift=sh, act=1, residual=res, res_f32=res_f32, out_f32=True)
    np.testing.assert_allclose(pre, np.maximum(base * sc + sh + res, 0), rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("shape", [(1, 64, 256), (3, 64, 128), (1, 96, 40)])
@pytest.mark.parametrize("res_f32", [False, True])
def test_conv_residual_upsampled(env, shape, res_f32):
    """out = conv + bias + nearest_upsample_x2(residual): the top-down Add of the FPN blocks
    (segmentation_models FPNBlock, /root/reference utils/model.py:200-211) in the epilogue."""
    k, cin, cout = shape
    rng = np.random.default_rng(cin + cout + k)
    n, h, w = 2, 12, 38
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (k, k, cin, cout)))
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    res = rng.normal(0, 1, (n, h // 2, w // 2, cout)).astype(np.float32)
    if not res_f32:
        res = bf16_round(res)
    want = ref_conv(x, wt, 1, k // 2) + sh + np.repeat(np.repeat(res, 2, axis=1), 2, axis=2)
    for out_f32 in (False, True):
        got, _ = run_conv(env, x, wt, 1, k // 2, shift=sh, residual=res, res_f32=res_f32, out_f32=out_f32, res_shift=1)
        tol = 1e-4 if out_f32 else 1e-2
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())


@pytest.mark.parametrize("shape", [(2, 11, 37, 64, 40), (1, 16, 24, 128, 256), (3, 9, 21, 192, 200), (1, 13, 50, 256, 640),
                                   (2, 30, 34, 64, 128)])
@pytest.mark.parametrize("mode", ["relu", "res", "res_after", "res_up", "out2", "out2_linear"])
def test_conv_pointwise_short_contraction(env, shape, mode):
    """1x1 layers with Cin in {64, 128, 192, 256} and 16-bit output run on the persistent point-wise kernel
    (weights of a 128-cout tile in LDS, pixel fragments straight from memory): LinkNet's decoder 1x1 layers with
    their skip additions, FPN's lateral layers with the up-sampled addend, both as channel slices of wider
    buffers (segmentation_models DecoderBlock / FPNBlock, /root/reference utils/model.py:188-211)."""
    L, lib, dev = env
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(cin * 7 + cout + len(mode))
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (1, 1, cin, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    base = ref_conv(x, wt, 1, 0) * sc + sh
    kw = dict(scale=sc, shift=sh, in_extra_c=8, out_extra_c=16)
    want2 = None
    if mode == "relu":
        want = np.maximum(base, 0)
        kw.update(act=1)
    elif mode in ("res", "res_after"):
        res = bf16_round(rng.normal(0, 1, (n, h, w, cout)))
        want = np.maximum(base + res, 0) if mode == "res" else np.maximum(base, 0) + res
        kw.update(act=1, residual=res, res_f32=False, res_after=mode == "res_after")
    elif mode == "res_up":
        if h % 2 or w % 2:
            h2, w2 = (h + 1) // 2, (w + 1) // 2
        else:
            h2, w2 = h // 2, w // 2
        res = bf16_round(rng.normal(0, 1, (n, h2, w2, cout)))
        want = base + np.repeat(np.repeat(res, 2, axis=1), 2, axis=2)[:, :h, :w]
        kw.update(residual=res, res_f32=False, res_shift=1)
    else:
        s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        t2 = rng.normal(0, 0.5, cout).astype(np.float32)
        want = base
        want2 = base * s2 + t2 if mode == "out2_linear" else np.maximum(base * s2 + t2, 0)
        kw.update(scale2=s2, shift2=t2, out2_linear=mode == "out2_linear")
    got, got2 = run_conv(env, x, wt, 1, 0, **kw)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-2 * np.abs(want).max())
    if want2 is not None:
        np.testing.assert_allclose(got2, want2, rtol=1e-2, atol=1e-2 * np.abs(want2).max())
    # the tiled kernels (any tile hint) give the same values up to the order of the fp32 sums
    alt, alt2 = run_conv(env, x, wt, 1, 0, tile_hint=1, **kw)
    np.testing.assert_allclose(got, alt, rtol=1e-2, atol=1e-2 * np.abs(want).max())
    # and the layer did run on the point-wise kernel
    d = new_desc(L)
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
    d.KH = d.KW = d.stride_y = d.stride_x = 1
    keep = torch.zeros(64, dtype=torch.float32, device=dev)
    d.inp = d.weight = d.out = keep.data_ptr()
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin + 8, (cin + 8) * w, (cin + 8) * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout + 16, (cout + 16) * w, (cout + 16) * w * h
    d.out_dtype = L.DT_BF16
    buf = C.create_string_buffer(128)
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode().startswith("conv_pw_thin_kernel<%d, 2>" % (cin // 16)), buf.value


def test_conv_epilogues(env):
    rng = np.random.default_rng(7)
    n, h, w, cin, cout = 2, 10, 12, 64, 64
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (3, 3, cin, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    base = ref_conv(x, wt, 1, 1)
    # scale/shift/relu, bf16 output inside a wider buffer (concat slice)
    got, _ = run_conv(env, x, wt, 1, 1, scale=sc, shift=sh, act=1, in_extra_c=32, out_extra_c=64)
    want = np.maximum(base * sc + sh, 0)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-2)   # bf16 store: 2^-8 relative
    # residual (fp32) + raw fp32 output + activated second output
    res = rng.normal(0, 1, (n, h, w, cout)).astype(np.float32)
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    t2 = rng.normal(0, 0.5, cout).astype(np.float32)
    got, got2 = run_conv(env, x, wt, 1, 1, residual=res, scale2=s2, shift2=t2, out_f32=True)
    np.testing.assert_allclose(got, base + res, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got2, np.maximum((base + res) * s2 + t2, 0), rtol=1e-2, atol=1e-2)
    # second output only, bf16 residual
    resb = bf16_round(res)
    _, got2 = run_conv(env, x, wt, 1, 1, residual=resb, res_f32=False, scale2=s2, shift2=t2,
                       want_out=False)
    np.testing.assert_allclose(got2, np.maximum((base + resb) * s2 + t2, 0), rtol=1e-2, atol=1e-2)
    # the second output without its ReLU (a biased copy of the result), through the LDS epilogue and, on a
    # 128-channel 1x1 layer with an up-sampled residual (FPN's p3 layer), through the direct one
    got, got2 = run_conv(env, x, wt, 1, 1, shift=sh, residual=resb, res_f32=False, scale2=s2, shift2=t2,
                         out2_linear=True)
    np.testing.assert_allclose(got, base + sh + resb, rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(got2, (base + sh + resb) * s2 + t2, rtol=1e-2, atol=3e-2)
    assert (got2 < 0).any()
    x1 = bf16_round(rng.normal(0, 1, (1, 16, 24, 64)))
    w1 = bf16_round(rng.normal(0, 0.1, (1, 1, 64, 128)))
    r1 = bf16_round(rng.normal(0, 1, (1, 8, 12, 128)))
    b2 = rng.normal(0, 0.5, 128).astype(np.float32)
    got, got2 = run_conv(env, x1, w1, 1, 0, residual=r1, res_f32=False, res_shift=1, scale2=np.ones(128, np.float32),
                         shift2=b2, out2_linear=True)
    want = ref_conv(x1, w1, 1, 0) + np.repeat(np.repeat(r1, 2, 1), 2, 2)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(got2, want + b2, rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("shape", [(3, 250, 400, 32, 32), (1, 37, 70, 32, 24), (2, 8, 32, 32, 8), (5, 129, 257, 32, 16)])
def test_conv_c32_persistent(env, shape):
    """Cin = 32, Cout <= 32 with bf16 output runs on persistent blocks (weights in registers,
    double-buffered halo, several patches per block); it accumulates the taps in the same order as
    the one-patch kernel (hint 128): the fp32 sums are equal, and the stored bf16 values may differ
    by one ulp on a few elements in 10^5 only where the two epilogues round scale * x + shift
    differently (fused against separate multiply and add).  Both agree with the oracle."""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(41)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (3, 3, cin, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    base = ref_conv(x, wt, 1, 1)
    for kw in (dict(), dict(scale=sc, shift=sh, act=1), dict(shift=sh, in_extra_c=32, out_extra_c=32)):
        got, _ = run_conv(env, x, wt, 1, 1, **kw)
        old, _ = run_conv(env, x, wt, 1, 1, tile_hint=128, **kw)
        if "scale" in kw:
            assert (got != old).mean() < flips(1e-4)
            np.testing.assert_allclose(got, old, rtol=1e-2, atol=1e-6)
        else:
            np.testing.assert_array_equal(got, old)
        want = base * kw.get("scale", 1.0) + kw.get("shift", 0.0)
        if kw.get("act"):
            want = np.maximum(want, 0)
        np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("shape", [(2, 300, 520, 64), (1, 38, 70, 64), (1, 16, 64, 40)])
def test_conv_stem_persistent(env, shape):
    """The 7x7 stride-2 stem in the plan builder's form (7x1, vertical stride 2, windows of
    8 pixels x 4 channels every 2 pixels of a zero-padded input) on the persistent stem kernel:
    against the generic kernel (hint 128) and against the oracle's 7x7 stride-2 convolution."""
    from oracle import ops_np
    L, lib, dev = env
    n, h, w, cout = shape
    rng = np.random.default_rng(53)
    x = bf16_round(rng.normal(0, 1, (n, h, w, 3)))
    k = bf16_round(rng.normal(0, 0.1, (7, 7, 3, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.5, cout).astype(np.float32)
    want = np.maximum(ops_np.conv2d(x, k, stride=2, pad=3) * sc + sh, 0)
    padded = np.zeros((n, h + 6, w + 8, 4), np.float32)
    padded[:, 3:3 + h, 3:3 + w, :3] = x
    wk = np.zeros((cout, 7, 1, 8, 4), np.float32)
    wk[:, :, 0, :7, :3] = k.transpose(3, 0, 1, 2)
    xin, wt = dev_bf16(padded, dev), dev_bf16(wk.reshape(cout, 7, 1, 32), dev)
    scd, shd = dev_f32(sc, dev), dev_f32(sh, dev)
    outs = []
    for hint in (0, 128):
        out = torch.full((n, h // 2, w // 2, cout), 7.0, dtype=h16(), device=dev)
        d = new_desc(L)
        d.inp, d.weight, d.out = xin.data_ptr(), wt.data_ptr(), out.data_ptr()
        d.N, d.H, d.W, d.Cin = n, h + 6, w // 2, 32
        d.Ho, d.Wo, d.Cout = h // 2, w // 2, cout
        d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = 7, 1, 2, 1, 0, 0
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = 8, (w + 8) * 4, (h + 6) * (w + 8) * 4
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * (w // 2), cout * (w // 2) * (h // 2)
        d.scale, d.shift, d.act, d.out_dtype, d.temperature = scd.data_ptr(), shd.data_ptr(), 1, L.DT_BF16, 1.0
        d.tile_hint = hint
        L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
        torch.cuda.synchronize()
        outs.append(out.float().cpu().numpy())
    np.testing.assert_allclose(outs[0], want, rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-2, atol=1e-2)   # summation order may differ
    assert (outs[0] != outs[1]).mean() < flips(0.02)


@pytest.mark.parametrize("k,pad", [(1, 0), (3, 1)])
def test_conv_prologue(env, k, pad):
    """BatchNorm+ReLU applied to the operand; padded taps must stay zero."""
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 1, 9, 11, 96, 128
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 0.1, (k, k, cin, cout)))
    s = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    t = rng.normal(0.3, 0.5, cin).astype(np.float32)    # positive shift: relu(shift) != 0
    got, _ = run_conv(env, x, wt, 1, pad, in_scale=s, in_shift=t, out_f32=True)
    want = ref_conv(x, wt, 1, pad, in_scale=s, in_shift=t)
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3 * np.abs(want).max())


@pytest.mark.parametrize("shape", [(1, 9, 11, 96, 128), (2, 16, 24, 256, 128), (1, 13, 17, 64, 64),
                                   (1, 8, 40, 1280, 640), (1, 20, 20, 160, 192), (3, 7, 9, 512, 72)])
@pytest.mark.parametrize("epilogue", [False, True])
def test_conv_pointwise_prologue_16bit(env, shape, epilogue):
    """1x1 convolution with BatchNorm+ReLU on its operand and 16-bit output (DenseNet's transition layers):
    the LDS-DMA kernel that applies the prologue to the fragments (hint 0) against the register-staged generic
    kernel (hint 1) and the oracle; ragged pixel and cout tiles, input and output as channel slices."""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(29)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    wt = bf16_round(rng.normal(0, 1.0 / np.sqrt(cin), (1, 1, cin, cout)))
    s = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    t = rng.normal(0.3, 0.5, cin).astype(np.float32)
    es = rng.uniform(0.5, 1.5, cout).astype(np.float32) if epilogue else None
    et = rng.normal(0, 0.3, cout).astype(np.float32) if epilogue else None
    act = 1 if epilogue else 0
    outs = []
    for hint in (0, 1):
        got, _ = run_conv(env, x, wt, 1, 0, scale=es, shift=et, act=act, in_scale=s, in_shift=t, tile_hint=hint,
                          in_extra_c=32, out_extra_c=32)
        outs.append(got)
    want = ref_conv(x, wt, 1, 0, in_scale=s, in_shift=t)
    if epilogue:
        want = np.maximum(want * es + et, 0)
    # one rounding of the stored result, and (fp16 storage) the prologue in packed fp16 arithmetic
    tol = 2 ** -7 if STORAGE["fmt"] == 0 else 2 ** -8
    np.testing.assert_allclose(outs[0], want, rtol=tol, atol=tol * np.abs(want).max())
    assert np.linalg.norm(outs[0] - want) <= 0.6 * tol * np.linalg.norm(want)
    np.testing.assert_allclose(outs[0], outs[1], rtol=tol, atol=tol * np.abs(want).max())
    L, lib, dev = env
    # the dispatch really took the new kernel for these shapes
    d = new_desc(L)
    xin = torch.zeros((n, h, w, cin), dtype=h16(), device=dev)
    wd = torch.zeros((cout, cin), dtype=h16(), device=dev)
    o = torch.zeros((n, h, w, cout), dtype=h16(), device=dev)
    sd, td = dev_f32(s, dev), dev_f32(t, dev)
    d.inp, d.weight, d.out, d.in_scale, d.in_shift = xin.data_ptr(), wd.data_ptr(), o.data_ptr(), sd.data_ptr(), td.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
    d.KH = d.KW = d.stride_y = d.stride_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * w, cout * w * h
    d.temperature = 1.0
    buf = C.create_string_buffer(128)
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode() == "conv_pw_pro_kernel", buf.value


@pytest.mark.parametrize("shape", [(2, 6, 7, 64, 32), (1, 9, 35, 128, 64), (1, 8, 32, 64, 160),
                                   (2, 5, 5, 32, 16), (1, 17, 40, 192, 128)])
@pytest.mark.parametrize("hint", [0, 11, 12, 15, 16, 32])
def test_conv_transpose_4x4s2(env, shape, hint):
    """Conv2DTranspose(4, 2, 'same') as four 2x2 phase convolutions; hint 16 = register-staged
    path, 32 = LDS-DMA path without the resident halo, 11/12/15 = halo variants."""
    from oracle import ops_np, ops_torch
    L, lib, dev = env
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(3)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    k = bf16_round(rng.normal(0, 0.1, (4, 4, cout, cin)))      # Keras layout
    want = ops_np.conv2d_transpose_4x4s2(x, k)
    want_t = ops_torch.to_numpy(ops_torch.conv2d_transpose_4x4s2(ops_torch.asarray(x), k))
    np.testing.assert_allclose(want, want_t, rtol=1e-4, atol=1e-4)
    ph = np.zeros((2, 2, cout, 2, 2, cin), np.float32)
    for a in range(2):
        for b in range(2):
            for dy in range(2):
                for dx in range(2):
                    ph[a, b, :, dy, dx, :] = k[3 - a - 2 * dy, 3 - b - 2 * dx]
    xin, wt = dev_bf16(x, dev), dev_bf16(ph, dev)
    out = torch.zeros((n, 2 * h, 2 * w, cout), dtype=torch.float32, device=dev)
    d = new_desc(L)
    d.inp, d.weight, d.out = xin.data_ptr(), wt.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
    d.KH = d.KW = 2
    d.stride_y = d.stride_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * 2 * w, cout * 4 * w * h
    d.out_dtype, d.transposed, d.temperature = L.DT_F32, 1, 1.0
    d.tile_hint = hint
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("shape", [(3, 100, 300, 64, 32), (2, 77, 250, 64, 24), (1, 8, 32, 64, 32)])
def test_conv_transpose_c64_persistent(env, shape):
    """64 -> (<= 32) channels: one block per CU walks several patches with the next halo in
    flight; more patches than CUs, ragged borders and a ragged channel count."""
    from oracle import ops_torch
    L, lib, dev = env
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(23)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    k = bf16_round(rng.normal(0, 0.1, (4, 4, cout, cin)))
    want = ops_torch.to_numpy(ops_torch.conv2d_transpose_4x4s2(ops_torch.asarray(x), k))
    ph = np.zeros((2, 2, cout, 2, 2, cin), np.float32)
    for a in range(2):
        for b in range(2):
            for dy in range(2):
                for dx in range(2):
                    ph[a, b, :, dy, dx, :] = k[3 - a - 2 * dy, 3 - b - 2 * dx]
    xin, wt = dev_bf16(x, dev), dev_bf16(ph, dev)
    for dt, tdt in ((L.DT_F32, torch.float32), (L.DT_BF16, h16())):
        out = torch.zeros((n, 2 * h, 2 * w, cout), dtype=tdt, device=dev)
        d = new_desc(L)
        d.inp, d.weight, d.out = xin.data_ptr(), wt.data_ptr(), out.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, cout
        d.KH = d.KW = 2
        d.stride_y = d.stride_x = 1
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * 2 * w, cout * 4 * w * h
        d.out_dtype, d.transposed, d.temperature = dt, 1, 1.0
        buf = C.create_string_buffer(128)
        L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
        # 16-bit output leaves through the direct epilogue: the 8-wave form; fp32 output keeps the 4-wave form
        assert buf.value.decode() == ("convT_c64_kernel" if dt == L.DT_F32 else "convT_c64w8_kernel"), buf.value
        L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        if dt == L.DT_F32:
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-4)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape", [(2, 9, 21, 64, 128), (1, 20, 40, 128, 200), (1, 8, 32, 192, 64),
                                   (1, 13, 50, 64, 72), (3, 40, 70, 128, 512)])
def test_conv_transpose_data_gradient(env, shape):
    """transposed = MISEG_CONV_DGRAD_T: the data gradient of Conv2DTranspose(4, 2, 'same') in one
    launch (four phase lattices of dY accumulated in the chunk loop) equals the 4x4 stride-2
    pad-1 convolution of dY with the layer's kernel."""
    L, lib, dev = env
    n, h, w, c, ci = shape          # dY: (n, 2h, 2w, c) -> dX: (n, h, w, ci)
    rng = np.random.default_rng(61)
    dy = bf16_round(rng.normal(0, 1, (n, 2 * h, 2 * w, c)))
    k = bf16_round(rng.normal(0, 0.05, (4, 4, c, ci)))            # Keras layout (kh, kw, Cout_T, Cin_T)
    want = ref_conv(dy, k, 2, 1)
    ph = np.zeros((2, 2, ci, 2, 2, c), np.float32)
    for a in range(2):
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    ph[a, b, :, ty, tx, :] = k[3 - a - 2 * ty, 3 - b - 2 * tx].T
    xin, wt = dev_bf16(dy, dev), dev_bf16(ph, dev)
    res = bf16_round(rng.normal(0, 1, (n, h, w, ci)))
    rest = dev_bf16(res, dev)
    for dt, tdt, with_res in ((L.DT_F32, torch.float32, False), (L.DT_BF16, h16(), True)):
        out = torch.zeros((n, h, w, ci), dtype=tdt, device=dev)
        d = new_desc(L)
        d.inp, d.weight, d.out = xin.data_ptr(), wt.data_ptr(), out.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, c, h, w, ci
        d.KH = d.KW = 2
        d.stride_y = d.stride_x = 1
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = c, c * 2 * w, c * 4 * w * h
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = ci, ci * w, ci * w * h
        d.out_dtype, d.transposed, d.temperature = dt, 2, 1.0
        if with_res:
            d.residual, d.res_dtype = rest.data_ptr(), L.DT_BF16
            d.res_pix_stride, d.res_row_stride, d.res_img_stride = ci, ci * w, ci * w * h
        L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        if with_res:
            np.testing.assert_allclose(got, want + res, rtol=1e-2, atol=3e-2)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4 * np.abs(want).max())


@pytest.mark.parametrize("classes,act", [(3, 2), (4, 2), (1, 3), (2, 2)])
def test_conv_head_softmax_collapse(env, classes, act):
    """3x3 head + bias + softmax/sigmoid + strict-argmax collapse in the epilogue."""
    from oracle import ops_np, post
    L, lib, dev = env
    rng = np.random.default_rng(5)
    n, h, w, cin = 2, 10, 12, 32
    x = bf16_round(rng.normal(0, 1, (n, h, w, cin)))
    k = bf16_round(rng.normal(0, 0.2, (3, 3, cin, classes)))
    bias = rng.normal(0, 0.1, classes).astype(np.float32)
    logits = ops_np.conv2d(x, k, pad=1, bias=bias)
    temp = 0.5
    want = ops_np.softmax(logits, temp) if act == 2 else ops_np.sigmoid(logits)
    xin, wt = dev_bf16(x, dev), dev_bf16(k.transpose(3, 0, 1, 2), dev)
    out = torch.zeros((n, h, w, classes), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, h, w, classes), dtype=torch.uint8, device=dev)
    b = dev_f32(bias, dev)
    d = new_desc(L)
    d.inp, d.weight, d.out, d.shift = xin.data_ptr(), wt.data_ptr(), out.data_ptr(), b.data_ptr()
    d.collapse_out = mask.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, h, w, classes
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = classes, classes * w, classes * w * h
    d.out_dtype, d.act, d.temperature, d.collapse_value = L.DT_F32, act, temp, 127
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)
    # the mask must be the reference collapse of the probabilities the kernel itself produced
    want_mask = np.stack([post.collapse_probabilities(got[i], 127) for i in range(n)])
    np.testing.assert_array_equal(mask.cpu().numpy(), want_mask)


@pytest.mark.parametrize("classes,act,hw", [(3, 2, (10, 12)), (3, 2, (9, 35)), (4, 2, (17, 33)), (1, 3, (8, 32)),
                                            (2, 2, (3, 70))])
def test_head_gather(env, classes, act, hw):
    """final_upsampling (bilinear x2) + 3x3 head from the nine taps' channel contractions done at half
    resolution: logits of the plain order of operations (oracle), ragged tiles included."""
    from oracle import ops_np, post
    L, lib, dev = env
    rng = np.random.default_rng(17)
    n, (h, w), cin = 2, hw, 16
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)
    k = rng.normal(0, 0.3, (3, 3, cin, classes)).astype(np.float32)
    bias = rng.normal(0, 0.1, classes).astype(np.float32)
    temp = 0.7
    logits = ops_np.conv2d(ops_np.upsample_bilinear2x(x), k, pad=1, bias=bias)
    want = ops_np.softmax(logits, temp) if act == 2 else ops_np.sigmoid(logits)
    ypix = (9 * classes + 31) // 32 * 32
    y = np.full((n, h, w, ypix), 7.0, np.float32)             # the padding channels are never read
    y[..., :9 * classes] = np.einsum("nhwc,tco->nhwto", x.astype(np.float64),
                                     k.reshape(9, cin, classes).astype(np.float64)).reshape(n, h, w, 9 * classes)
    yd, bd = dev_f32(y, dev), dev_f32(bias, dev)
    out = torch.zeros((n, 2 * h, 2 * w, classes), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, 2 * h, 2 * w, classes), dtype=torch.uint8, device=dev)
    L.check(lib.miseg_head_gather_fwd(yd.data_ptr(), bd.data_ptr(), out.data_ptr(), mask.data_ptr(),
                                             n, h, w, classes, ypix, act, temp, 127, 0, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)
    want_mask = np.stack([post.collapse_probabilities(got[i], 127) for i in range(n)])
    np.testing.assert_array_equal(mask.cpu().numpy(), want_mask)
    # no mask asked for: the probabilities alone
    out2 = torch.zeros_like(out)
    L.check(lib.miseg_head_gather_fwd(yd.data_ptr(), bd.data_ptr(), out2.data_ptr(), None,
                                             n, h, w, classes, ypix, act, temp, 127, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    assert lib.miseg_head_gather_fwd(yd.data_ptr(), bd.data_ptr(), out2.data_ptr(), None,
                                            n, h, w, 5, ypix, act, temp, 127, 0, None) != 0


@pytest.mark.parametrize("classes,hw", [(3, (10, 12)), (3, (9, 35)), (1, (8, 32)), (2, (3, 7)), (4, (5, 6))])
def test_head_gather_backward_is_the_adjoint(env, classes, hw):
    """miseg_head_gather_bwd against torch autograd through UpSampling2D(bilinear) + the tap sum, in the
    class-major channel order the training path uses; and the forward in that order."""
    import torch.nn.functional as F
    L, lib, dev = env
    if STORAGE["suffix"]:
        pytest.skip("training tensors are bf16")
    rng = np.random.default_rng(23)
    n, (h, w) = 2, hw
    ypix = 32 if classes < 4 else 64
    y = torch.from_numpy(rng.normal(0, 1, (n, h, w, 9 * classes)).astype(np.float32)).requires_grad_(True)
    # logits[c] = sum_tap shift_tap(upsample(y[c * 9 + tap])), zero outside the image
    up = F.interpolate(y.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    pad = F.pad(up, (1, 1, 1, 1))
    H, W = 2 * h, 2 * w
    logits = torch.stack([sum(pad[:, c * 9 + ky * 3 + kx, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3))
                          for c in range(classes)], dim=-1)
    dz = bf16_round(rng.normal(0, 1, (n, H, W, classes)))
    logits.backward(torch.from_numpy(dz))
    want = y.grad.numpy()
    dz8 = np.full((n, H, W, 8), 3.0, np.float32)             # channels past `classes` are never read
    dz8[..., :classes] = dz
    dzd = dev_bf16(dz8, dev)
    dy = torch.full((n, h, w, ypix), 9.0, dtype=torch.bfloat16, device=dev)
    L.check(lib.miseg_head_gather_bwd(dzd.data_ptr(), 8, dy.data_ptr(), n, h, w, classes, ypix, None))
    torch.cuda.synchronize()
    got = dy.float().cpu().numpy()
    assert np.abs(got[..., 9 * classes:]).max() == 0
    np.testing.assert_allclose(got[..., :9 * classes], want, rtol=2 ** -7, atol=2e-2)
    assert np.linalg.norm(got[..., :9 * classes] - want) <= 3e-3 * np.linalg.norm(want)
    # forward, class-major: the same logits
    yp = np.zeros((n, h, w, ypix), np.float32)
    yp[..., :9 * classes] = y.detach().numpy()
    out = torch.zeros((n, H, W, classes), dtype=torch.float32, device=dev)
    act = 2 if classes > 1 else 3
    L.check(lib.miseg_head_gather_fwd(dev_f32(yp, dev).data_ptr(), None, out.data_ptr(), None, n, h, w, classes,
                                      ypix, act, 1.0, 0, 1, None))
    torch.cuda.synchronize()
    lg = logits.detach()
    ref = torch.softmax(lg, dim=-1) if classes > 1 else torch.sigmoid(lg)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 17, 40, 8, 32), (1, 8, 32, 8, 16), (1, 9, 33, 16, 32)])
def test_conv_narrow_input(env, shape):
    """in_valid_c: a 3x3 layer with Cin = 32 whose input pixels hold only the first 8 (16) channels in memory
    (pixel stride = that count); the missing channels read as zero, whatever the weights there."""
    from oracle import ops_np
    L, lib, dev = env
    n, h, w, cv, cout = shape
    rng = np.random.default_rng(31)
    x = bf16_round(rng.normal(0, 1, (n, h, w, cv)))
    k = bf16_round(rng.normal(0, 0.2, (3, 3, 32, cout)))          # taps for all 32 channels, 24 of them unused
    want = ops_np.conv2d(x, k[:, :, :cv], pad=1)
    xin, wt = dev_bf16(x, dev), dev_bf16(k.transpose(3, 0, 1, 2), dev)
    out = torch.zeros((n, h, w, cout), dtype=h16(), device=dev)
    d = new_desc(L)
    d.inp, d.weight, d.out = xin.data_ptr(), wt.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, 32, h, w, cout
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = cv, cv * w, cv * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * w, cout * w * h
    d.temperature, d.in_valid_c = 1.0, cv
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2 ** -7, atol=2e-2)
    assert np.linalg.norm(got - want) <= 4e-3 * np.linalg.norm(want)
    # shapes no kernel reads that way are refused, not computed on memory that does not exist
    d.KH = d.KW = 1
    d.pad_y = d.pad_x = 0
    assert lib.miseg_conv2d_fwd(C.byref(d), None) != 0
    d.KH = d.KW = 3
    d.pad_y = d.pad_x = 1
    d.in_valid_c = 12
    assert lib.miseg_conv2d_fwd(C.byref(d), None) != 0


def test_prep_input_and_stem(env):
    """bn_data + ZeroPadding2D(3) + 7x7/2 stem conv via the channel-padded 7x1 form."""
    from oracle import ops_np
    L, lib, dev = env
    rng = np.random.default_rng(9)
    n, h, w = 2, 32, 64
    x = rng.uniform(0, 1, (n, h, w, 3)).astype(np.float32)
    s = rng.uniform(0.5, 2, 3).astype(np.float32)
    t = rng.normal(0, 0.3, 3).astype(np.float32)
    k = bf16_round(rng.normal(0, 0.1, (7, 7, 3, 64)))
    xd = dev_f32(x, dev)
    padded = torch.full((n, h + 6, w + 8, 4), 5.0, dtype=h16(), device=dev)
    sd, td = dev_f32(s, dev), dev_f32(t, dev)
    L.check(fn(lib, "miseg_prep_input")(xd.data_ptr(), padded.data_ptr(), sd.data_ptr(), td.data_ptr(),
                                 n, h, w, None))
    torch.cuda.synchronize()
    p = padded.float().cpu().numpy()
    # the kernel contracts x*s+t into one fma, numpy rounds twice: allow one bf16 ulp
    np.testing.assert_allclose(p[:, 3:3 + h, 3:3 + w, :3], bf16_round(x * s + t), rtol=2 ** -7, atol=1e-6)
    xb = p[:, 3:3 + h, 3:3 + w, :3].copy()
    assert p[:, :3].max() == 0 and p[:, 3 + h:].max() == 0 and p[..., 3].max() == 0
    assert np.abs(p[:, :, :3]).max() == 0 and np.abs(p[:, :, 3 + w:]).max() == 0
    wk = np.zeros((64, 7, 1, 8, 4), np.float32)
    wk[:, :, 0, :7, :3] = k.transpose(3, 0, 1, 2)
    wt = dev_bf16(wk.reshape(64, 7, 1, 32), dev)
    out = torch.zeros((n, h // 2, w // 2, 64), dtype=torch.float32, device=dev)
    d = new_desc(L)
    d.inp, d.weight, d.out = padded.data_ptr(), wt.data_ptr(), out.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h + 6, w // 2, 32, h // 2, w // 2, 64
    d.KH, d.KW, d.stride_y, d.stride_x = 7, 1, 2, 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 8, (w + 8) * 4, (h + 6) * (w + 8) * 4
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 64, 64 * (w // 2), 64 * (w // 2) * (h // 2)
    d.out_dtype, d.temperature = L.DT_F32, 1.0
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
    torch.cuda.synchronize()
    want = ops_np.conv2d(xb, k, stride=2, pad=3)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_pointwise_kernels(env):
    from oracle import ops_np
    L, lib, dev = env
    rng = np.random.default_rng(13)
    n, h, w, c = 2, 12, 20, 40
    x = bf16_round(np.abs(rng.normal(0, 1, (n, h, w, c))))
    xd = dev_bf16(x, dev)
    # max pool (+ scale/shift/relu)
    s = rng.uniform(0.5, 1.5, c).astype(np.float32)
    t = rng.normal(0, 0.5, c).astype(np.float32)
    out = torch.zeros((n, h // 2, w // 2, c), dtype=h16(), device=dev)
    L.check(fn(lib, "miseg_maxpool3x3s2_fwd")(xd.data_ptr(), out.data_ptr(), None, None, n, h, w, c, c, c, None))
    np.testing.assert_array_equal(out.float().cpu().numpy(), ops_np.maxpool3x3s2_pad1(x))
    sd, td = dev_f32(s, dev), dev_f32(t, dev)
    L.check(fn(lib, "miseg_maxpool3x3s2_fwd")(xd.data_ptr(), out.data_ptr(), sd.data_ptr(), td.data_ptr(),
                                       n, h, w, c, c, c, None))
    got, want = out.float().cpu().numpy(), bf16_round(np.maximum(ops_np.maxpool3x3s2_pad1(x) * s + t, 0))
    np.testing.assert_allclose(got, want, rtol=2 ** -7, atol=1e-6)
    assert (got != want).mean() < flips(1e-4)
    # average pool
    L.check(fn(lib, "miseg_avgpool2x2_fwd")(xd.data_ptr(), out.data_ptr(), n, h, w, c, c, c, None))
    np.testing.assert_allclose(out.float().cpu().numpy(), ops_np.avgpool2x2(x), rtol=1e-2, atol=1e-3)
    # nearest upsample x4, and x2 + add
    up = torch.zeros((n, 4 * h, 4 * w, c), dtype=h16(), device=dev)
    L.check(fn(lib, "miseg_upsample_nearest_fwd")(xd.data_ptr(), None, up.data_ptr(), n, h, w, c, 2, c, 0, c, None))
    np.testing.assert_array_equal(up.float().cpu().numpy(), ops_np.upsample_nearest(x, 4))
    y = bf16_round(rng.normal(0, 1, (n, 2 * h, 2 * w, c)))
    yd = dev_bf16(y, dev)
    L.check(fn(lib, "miseg_upsample_nearest_fwd")(xd.data_ptr(), yd.data_ptr(), yd.data_ptr(), n, h, w, c, 1, c, c, c, None))
    np.testing.assert_array_equal(yd.float().cpu().numpy(), bf16_round(ops_np.upsample_nearest(x, 2) + y))
    # bilinear x2
    up2 = torch.zeros((n, 2 * h, 2 * w, c), dtype=h16(), device=dev)
    L.check(fn(lib, "miseg_upsample_bilinear2x_fwd")(xd.data_ptr(), up2.data_ptr(), n, h, w, c, c, c, None))
    np.testing.assert_allclose(up2.float().cpu().numpy(), ops_np.upsample_bilinear2x(x), rtol=1e-2, atol=1e-3)
    # scale/shift/relu
    o = torch.zeros_like(xd)
    L.check(fn(lib, "miseg_scale_shift_relu_fwd")(xd.data_ptr(), o.data_ptr(), sd.data_ptr(), td.data_ptr(),
                                           n * h * w, c, c, c, None))
    # the kernel contracts x*s+t into one fma, numpy rounds the product first: one ulp on a few values
    got, want = o.float().cpu().numpy(), bf16_round(np.maximum(x * s + t, 0))
    np.testing.assert_allclose(got, want, rtol=2 ** -7, atol=1e-6)
    assert (got != want).mean() < flips(1e-4)
    torch.cuda.synchronize()


def test_vgg_layer_kernels(env):
    """The layers only the VGG encoders use: MaxPooling2D(2) and the 3x3 neighbourhood layout of the image
    that turns block1_conv1 into a 1x1 convolution; both exact."""
    from oracle import ops_np
    L, lib, dev = env
    rng = np.random.default_rng(31)
    n, h, w, c = 2, 12, 20, 40
    x = bf16_round(rng.normal(0, 1, (n, h, w, c)))
    # pooling out of, and into, channel slices of wider buffers
    wide = torch.zeros((n, h, w, c + 24), dtype=h16(), device=dev)
    wide[..., 8:8 + c] = dev_bf16(x, dev)
    out = torch.full((n, h // 2, w // 2, c + 8), 7.0, dtype=h16(), device=dev)
    L.check(lib.miseg_maxpool2x2_fwd(wide.data_ptr() + 16, out.data_ptr() + 16, n, h, w, c, c + 24, c + 8,
                                     STORAGE["fmt"], None))
    got = out.float().cpu().numpy()
    np.testing.assert_array_equal(got[..., 8:], ops_np.maxpool2x2(x))
    np.testing.assert_array_equal(got[..., :8], 7.0)
    assert lib.miseg_maxpool2x2_fwd(wide.data_ptr(), out.data_ptr(), n, h + 1, w, c, c + 24, c + 8,
                                    STORAGE["fmt"], None) < 0          # odd height
    # image neighbourhoods
    img = rng.random((n, h, w, 3)).astype(np.float32)
    col = torch.full((n, h, w, 32), 9.0, dtype=h16(), device=dev)
    L.check(lib.miseg_im2col3x3_fwd(dev_f32(img, dev).data_ptr(), col.data_ptr(), n, h, w, 9, STORAGE["fmt"], None))
    pad = np.pad(img, ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = np.zeros((n, h, w, 32), np.float32)
    for ky in range(3):
        for kx in range(3):
            want[..., (ky * 3 + kx) * 3:(ky * 3 + kx) * 3 + 3] = pad[:, ky:ky + h, kx:kx + w, :]
    np.testing.assert_array_equal(col.float().cpu().numpy(), bf16_round(want))
    # the 3x3 'same' convolution of the image == the 1x1 convolution of the neighbourhood tensor
    k = rng.normal(0, 0.2, (3, 3, 3, 64)).astype(np.float32)
    w1 = np.zeros((1, 1, 32, 64), np.float32)
    w1[0, 0, :27] = k.reshape(27, 64)
    bias = rng.normal(0, 0.1, 64).astype(np.float32)
    got, _ = run_conv(env, bf16_round(want), bf16_round(w1), shift=bias, act=L.ACT_RELU)
    ref = np.maximum(ref_conv(bf16_round(img), bf16_round(k), 1, 1) + bias, 0)
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2)
    L.check(lib.miseg_im2col3x3_fwd(dev_f32(img, dev).data_ptr(), col.data_ptr(), n, h, w, 1, STORAGE["fmt"], None))
    want1 = np.zeros((n, h, w, 32), np.float32)
    want1[..., :3] = img
    np.testing.assert_array_equal(col.float().cpu().numpy(), bf16_round(want1))
    torch.cuda.synchronize()


def test_collapse_and_loss_sums(env):
    from oracle import losses, post
    L, lib, dev = env
    rng = np.random.default_rng(17)
    h, w, c = 40, 50, 3
    p = rng.dirichlet(np.ones(c), size=(h, w)).astype(np.float32)
    p[0, 0] = (0.5, 0.5, 0.0)        # ties: the in-place loop gives (0,127,0)
    p[0, 1] = (0.25, 0.25, 0.5)
    p[0, 2] = (1 / 3, 1 / 3, 1 / 3)
    p[0, 3] = (0.4, 0.2, 0.4)
    pd = dev_f32(p, dev)
    out = torch.zeros((h, w, c), dtype=torch.uint8, device=dev)
    L.check(lib.miseg_collapse_u8(pd.data_ptr(), out.data_ptr(), h * w, c, 127, None))
    np.testing.assert_array_equal(out.cpu().numpy(), post.collapse_probabilities(p, 127))
    gt = post.one_hot_encode(rng.integers(0, c, (h, w, 1)), c)
    gd = dev_f32(gt, dev)
    sums = torch.zeros(4 * c, dtype=torch.float64, device=dev)
    L.check(lib.miseg_loss_sums(gd.data_ptr(), pd.data_ptr(), sums.data_ptr(), h * w, c, None))
    s = sums.cpu().numpy().reshape(4, c)
    tp, spr, sgt, sl = s
    f1 = np.mean((2 * tp + 1e-5) / (spr + sgt + 1e-5))
    iou = np.mean((tp + 1e-5) / (spr + sgt - tp + 1e-5))
    cce = -sl.sum() / (h * w * c)
    assert abs(f1 - losses.f1_score(gt[None], p[None])) < 1e-5
    assert abs(iou - losses.iou_score(gt[None], p[None])) < 1e-5
    assert abs(cce + 1 - f1 - losses.cce_dice_loss(gt[None], p[None])) < 1e-5


@pytest.mark.parametrize("shape", [(2, 19, 70, 64), (1, 8, 32, 64), (1, 37, 45, 40), (2, 16, 64, 48)])
@pytest.mark.parametrize("variant", ["plain", "res", "res_after", "res_out2", "out2_only", "linear2"])
def test_conv_c64p(env, shape, variant):
    """The persistent 64-channel kernel (3x3, Cin = 64, 32 < Cout <= 64: stage 1 of the ResNets built through
    /root/reference utils/model.py:176-186) against the oracle: ragged patches and cout groups, residual before
    and after the activation, second output (activated or linear), a launch that stores its second output only;
    and the dispatch really takes it, while hint 12 keeps the K-split ring kernel for comparison."""
    L, lib, dev = env
    n, h, w, cout = shape
    rng = np.random.default_rng(hash((shape, variant)) % 2**32)
    x = bf16_round(rng.normal(0, 1, (n, h, w, 64)))
    wt = bf16_round(rng.normal(0, 0.06, (3, 3, 64, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.3, cout).astype(np.float32)
    res = bf16_round(rng.normal(0, 1, (n, h, w, cout))) if variant.startswith("res") or variant == "out2_only" else None
    s2 = rng.uniform(0.5, 1.5, cout).astype(np.float32) if variant in ("res_out2", "out2_only", "linear2") else None
    t2 = rng.normal(0, 0.3, cout).astype(np.float32) if s2 is not None else None
    after = variant == "res_after"
    act = 1 if variant in ("plain", "res_after") else 0
    kw = dict(scale=sc, shift=sh, act=act, residual=res, res_f32=False, scale2=s2, shift2=t2, res_after=after,
              want_out=variant != "out2_only", out2_linear=variant == "linear2")
    got, got2 = run_conv(env, x, wt, 1, 1, **kw)
    v = ref_conv(x, wt, 1, 1) * sc + sh
    if res is not None and not after:
        v = v + res
    want = np.maximum(v, 0) if act else v
    if after:
        want = want + res
    tol = 2 ** -7 if STORAGE["fmt"] == 0 else 2 ** -8
    if got is not None:
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
    if s2 is not None:
        want2 = v * s2 + t2
        if variant != "linear2":
            want2 = np.maximum(want2, 0)
        np.testing.assert_allclose(got2, want2, rtol=tol, atol=2 * tol * np.abs(want2).max())
    # the same launch on the K-split ring kernel
    ring, ring2 = run_conv(env, x, wt, 1, 1, tile_hint=L.TILE_NO_PERSIST, **kw)
    if got is not None:
        np.testing.assert_allclose(got, ring, rtol=tol, atol=tol * np.abs(want).max())
    if s2 is not None:
        np.testing.assert_allclose(got2, ring2, rtol=tol, atol=2 * tol * np.abs(want).max())
    d = new_desc(L)
    t = torch.zeros((n, h, w, 64), dtype=h16(), device=dev)
    d.inp = d.weight = d.out = t.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, 64, h, w, cout
    d.KH = d.KW = 3
    d.stride_y = d.stride_x = d.pad_y = d.pad_x = 1
    d.in_pix_stride, d.in_row_stride, d.in_img_stride = 64, 64 * w, 64 * w * h
    d.out_pix_stride, d.out_row_stride, d.out_img_stride = 64, 64 * w, 64 * w * h
    d.out_dtype, d.temperature = L.DT_BF16, 1.0
    buf = C.create_string_buffer(128)
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode() == "conv3x3_c64p_kernel<false, false>", buf.value
    d.tile_hint = L.TILE_NO_PERSIST
    L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
    assert buf.value.decode().startswith("conv3x3_halo_ks_kernel"), buf.value


@pytest.mark.parametrize("shape", [(1, 5, 9, 64), (3, 8, 31, 56), (1, 70, 33, 64)])
def test_conv_c64p_channel_slices_and_small_images(env, shape):
    """The persistent 64-channel kernel on views: input and output as channel slices of wider tensors (what the
    engine's concatenation buffers are), images smaller than one 8 x 32 patch, more patches than blocks."""
    n, h, w, cout = shape
    rng = np.random.default_rng(hash(shape) % 2**32)
    x = bf16_round(rng.normal(0, 1, (n, h, w, 64)))
    wt = bf16_round(rng.normal(0, 0.06, (3, 3, 64, cout)))
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.normal(0, 0.3, cout).astype(np.float32)
    got, _ = run_conv(env, x, wt, 1, 1, scale=sc, shift=sh, act=1, in_extra_c=32, out_extra_c=64)
    want = np.maximum(ref_conv(x, wt, 1, 1) * sc + sh, 0)
    tol = 2 ** -7 if STORAGE["fmt"] == 0 else 2 ** -8
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
