"""Time one weight-gradient launch (debug aid)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd  # noqa: E402,F401
from miseg_amd import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--h", type=int, default=1920)
ap.add_argument("--w", type=int, default=2560)
ap.add_argument("--cin", type=int, default=32)
ap.add_argument("--cout", type=int, default=32)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--hints", default="1,2")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
lib = L.load()
pad = (a.k - 1) // 2
x = torch.randn((a.n, a.h, a.w, a.cin), device="cuda").to(torch.bfloat16)
dy = torch.randn((a.n, a.h, a.w, a.cout), device="cuda").to(torch.bfloat16)
dw = torch.zeros((a.cout, a.k, a.k, a.cin), device="cuda")
res = {}
for hint in map(int, a.hints.split(",")):
    d = L.WgradDesc()
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = a.n, a.h, a.w, a.cin, a.h, a.w, a.cout
    d.KH = d.KW = a.k
    d.stride_y = d.stride_x = 1
    d.pad_y = d.pad_x = pad
    d.x_pix, d.x_row, d.x_img = a.cin, a.cin * a.w, a.cin * a.w * a.h
    d.dy_pix, d.dy_row, d.dy_img = a.cout, a.cout * a.w, a.cout * a.w * a.h
    d.tile_hint = hint
    dw.zero_()
    L.check(lib.miseg_conv2d_wgrad(C.byref(d), None))
    torch.cuda.synchronize()
    res[hint] = dw.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        L.check(lib.miseg_conv2d_wgrad(C.byref(d), torch.cuda.current_stream().cuda_stream))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = 2.0 * a.n * a.h * a.w * a.k * a.k * a.cin * a.cout
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    print("hint %d: %.3f ms  %.1f TFLOP/s  %.2f TB/s (operands once)" % (hint, ms, fl / ms / 1e9, gb / ms))
if len(res) == 2:
    k = list(res)
    print("max rel diff between kernels: %.2e" % float((res[k[0]] - res[k[1]]).abs().max() / res[k[0]].abs().max()))
