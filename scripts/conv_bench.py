"""Time one convolution shape through the C ABI (for rocprofv3 --pmc runs and tile sweeps).
usage: conv_bench.py N H W Cin Cout K [stride] [--hint H] [--iters I] [--transposed]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import miseg_amd
from miseg_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("dims", type=int, nargs="+")
ap.add_argument("--hint", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--transposed", action="store_true")
ap.add_argument("--relu", action="store_true")
ap.add_argument("--softmax", action="store_true")
ap.add_argument("--f32out", action="store_true")
ap.add_argument("--residual", action="store_true", help="16-bit residual added before the activation")
ap.add_argument("--res-up", action="store_true", help="residual at half resolution, read at (y >> 1, x >> 1)")
ap.add_argument("--out2", action="store_true", help="second (linear) 16-bit output")
ap.add_argument("--res-after", action="store_true", help="... after the activation (LinkNet's skip additions)")
a = ap.parse_args()
n, h, w, cin, cout, k = a.dims[:6]
stride = a.dims[6] if len(a.dims) > 6 else 1
lib = L.load()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn((n, h, w, cin), generator=g).to(torch.bfloat16).to(dev)
if a.transposed:
    wt = (torch.randn((4, cout, 2, 2, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    ho, wo, oh, ow, kk, pad = h, w, 2 * h, 2 * w, 2, 0
else:
    wt = (torch.randn((cout, k, k, cin), generator=g) * 0.05).to(torch.bfloat16).to(dev)
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    oh, ow, kk = ho, wo, k
out = torch.zeros((n, oh, ow, cout), dtype=torch.bfloat16, device=dev)
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
d = L.ConvDesc()
d.inp, d.weight, d.out = x.data_ptr(), wt.data_ptr(), out.data_ptr()
d.scale, d.shift = sc.data_ptr(), sh.data_ptr()
d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
d.KH = d.KW = kk
d.stride_y = d.stride_x = 1 if a.transposed else stride
d.pad_y = d.pad_x = pad
d.in_pix_stride, d.in_row_stride, d.in_img_stride = cin, cin * w, cin * w * h
d.out_pix_stride, d.out_row_stride, d.out_img_stride = cout, cout * ow, cout * ow * oh
d.act = 1 if a.relu else 0
if a.softmax or a.f32out:
    out = torch.zeros((n, oh, ow, cout), dtype=torch.float32, device=dev)
    d.out = out.data_ptr()
    d.out_dtype = L.DT_F32
if a.softmax:
    mask = torch.zeros((n, oh, ow, cout), dtype=torch.uint8, device=dev)
    d.act = L.ACT_SOFTMAX
    d.collapse_out = mask.data_ptr()
    d.collapse_value = 127
if a.out2:
    out2 = torch.zeros((n, oh, ow, cout), dtype=torch.bfloat16, device=dev)
    d.out2 = out2.data_ptr()
    d.out2_pix_stride, d.out2_row_stride, d.out2_img_stride = cout, cout * ow, cout * ow * oh
    d.scale2, d.shift2 = sc.data_ptr(), sh.data_ptr()
    d.out2_linear = 1
if a.res_up:
    res = torch.randn((n, oh // 2, ow // 2, cout), generator=g).to(torch.bfloat16).to(dev)
    d.residual = res.data_ptr()
    d.res_pix_stride, d.res_row_stride, d.res_img_stride = cout, cout * (ow // 2), cout * (ow // 2) * (oh // 2)
    d.res_dtype = L.DT_BF16
    d.res_upsample_shift = 1
if a.residual or a.res_after:
    res = torch.randn((n, oh, ow, cout), generator=g).to(torch.bfloat16).to(dev)
    d.residual = res.data_ptr()
    d.res_pix_stride, d.res_row_stride, d.res_img_stride = cout, cout * ow, cout * ow * oh
    d.res_dtype = L.DT_BF16
    d.res_after_act = 1 if a.res_after else 0
d.transposed = 1 if a.transposed else 0
d.temperature = 1.0
d.tile_hint = a.hint
for _ in range(3):
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    L.check(lib.miseg_conv2d_fwd(C.byref(d), None))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
macs = n * ho * wo * cout * kk * kk * cin * (4 if a.transposed else 1)
print("dims %s hint %d: %.4f ms  %.1f TFLOP/s" % (a.dims, a.hint, ms, 2 * macs / ms / 1e9))
