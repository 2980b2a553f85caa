"""Lowering of a network graph (netspec.Graph) plus its weights to a recorded
plan of libmiseg.so kernel launches for inference on one MI355X, i.e. the
replacement of `model(batch, training=False)` at /root/reference
utils/predict.py:159.

Design (inference mode, BatchNorm with moving statistics):
  * activations and convolution weights are 16-bit NHWC in HBM — IEEE fp16 by default
    (`storage="fp16"`: 11 significand bits, which is what mask parity with an fp32 run takes),
    or bf16 (`storage="bf16"`, the format of the training path) — accumulation is fp32 on
    MFMA, the ResNet residual stream and the output probabilities are fp32;
  * every BatchNormalization+ReLU is folded either into the epilogue of the
    convolution that produces its input (scale/shift per output channel), into
    a second "activated" output of the convolution that closes a residual unit,
    or into the operand prologue of the consuming convolution (DenseNet, where
    one concatenated tensor feeds many differently normalised consumers);
  * Concatenate never copies: producers write channel slices of one buffer;
  * Conv2DTranspose(4, 2, 'same') runs as four 2x2 phase convolutions;
  * the 3x3 head convolution applies softmax (optionally temperature-scaled) or
    sigmoid and the strict-argmax collapse in its epilogue.
All buffers are allocated once when the plan is built; a forward pass is one
`miseg_plan_run` call.  torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .netspec import RESNET_UNITS, RESNET_BOTTLENECK, DENSENET_LAYERS, VGG_BLOCKS


class View:
    """Channel slice [c0, c0+C) of an NHWC device buffer."""

    def __init__(self, base, n, h, w, c, c0=0):
        self.base, self.n, self.h, self.w, self.c, self.c0 = base, n, h, w, c, c0
        self.ctot = base.shape[-1]
        self.esize = base.element_size()

    @property
    def ptr(self):
        return self.base.data_ptr() + self.c0 * self.esize

    @property
    def pix(self):
        return self.ctot

    @property
    def row(self):
        return self.ctot * self.w

    @property
    def img(self):
        return self.ctot * self.w * self.h

    def slice(self, c0, c):
        return View(self.base, self.n, self.h, self.w, c, self.c0 + c0)

    def torch(self):
        return self.base[..., self.c0:self.c0 + self.c]


def _ceil32(c):
    return (c + 31) // 32 * 32


class Engine:
    def __init__(self, graph, weights, batch=1, device="cuda:0", collapse_value=127,
                 keep_logits=False, fuse_tail=True, residual_bf16=None, storage="fp16", fuse_dense=True,
                 head_gather=True):
        """graph: netspec.Graph; weights: {layer: {weight_name: np.ndarray fp32}}."""
        if not torch.cuda.is_available():
            raise L.MisegError("no GPU visible: the HIP engine has no CPU fallback")
        self.lib = L.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        L.check(self.lib.miseg_device_check())
        self.g = graph
        self.w = weights
        self.n = batch
        # 16-bit storage format of activations and weights.  Both run the same kernels at the same
        # rate; with bf16 (8 significand bits) the strict-argmax masks of a random-init ResNet-34
        # U-Net agree with the fp32 network on 99.43 % of the pixels at 1920x2560, with fp16 on
        # more than 99.9 %.  fp16 saturates at 65504 (post-BatchNorm activations are far below it).
        if storage not in ("fp16", "bf16"):
            raise ValueError("storage must be 'fp16' or 'bf16', got %r" % (storage,))
        self.storage = storage
        self.fmt = L.FMT_F16 if storage == "fp16" else L.FMT_BF16
        self.tdtype = torch.float16 if storage == "fp16" else torch.bfloat16
        # U-Net: decoder_stage4b and final_conv as one kernel (the tensor between them stays on chip)
        self.fuse_tail = fuse_tail
        # DenseNet: each layer (BN-ReLU-1x1-BN-ReLU-3x3-concat) as one kernel, its 128-channel
        # intermediate tensor on chip
        self.fuse_dense = fuse_dense
        # FPN: final_stage as a 3x3 convolution of the finest tensor plus a phase (transpose-form) convolution of
        # the coarser ones at half resolution (see _fpn_final_stage_split); fuse_tail=False keeps the layer-wise form
        self.split_fpn_stage = fuse_tail
        # FPN: the head's channel contraction at half resolution + a gather kernel (_fpn_head_gather) instead of
        # the kernel that interpolates all 128 channels at full resolution (_fpn_head)
        self.head_gather = head_gather
        # ResNet residual stream in the 16-bit storage format instead of fp32: the convolutions that
        # close a residual unit move 4 instead of 10 bytes per element (+4.7 % images/s at
        # 1920x2560).  Measured mask agreement with the fp32 oracle at 1920x2560: fp16 storage
        # 99.932 % -> 99.928 % (still above the 99.9 % bar: on by default there); bf16 storage
        # 99.43 % -> 99.40 % (already below the bar: off by default, parity first).
        # None = that default; True / False force it.
        self.residual_bf16 = (storage == "fp16") if residual_bf16 is None else bool(residual_bf16)
        self.h, self.wd, _ = graph.input_shape
        self.classes = graph.classes
        self.collapse_value = collapse_value
        self._keep = []            # device tensors referenced by the plan
        self.taps = {}             # name -> View, for tests
        self.op_names = []
        self.op_bytes = []        # algorithmic HBM bytes of each conv op: operands read once, results written once
        self.plan = C.c_void_p(self.lib.miseg_plan_create())
        L.check(self.lib.miseg_plan_set_format(self.plan, self.fmt))
        self.x_in = torch.zeros((batch, self.h, self.wd, 3), dtype=torch.float32, device=self.device)
        self.probs = torch.zeros((batch, self.h, self.wd, self.classes), dtype=torch.float32,
                                 device=self.device)
        self.mask = torch.zeros((batch, self.h, self.wd, self.classes), dtype=torch.uint8,
                                device=self.device)
        self._lower()

    def __del__(self):
        try:
            if self.plan:
                self.lib.miseg_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # ------------------------------------------------------------------ memory
    def buf(self, h, w, c, dtype=None):
        dtype = dtype or self.tdtype
        t = torch.zeros((self.n, h, w, c), dtype=dtype, device=self.device)
        self._keep.append(t)
        return View(t, self.n, h, w, c)

    def dev_f32(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    def dev_h16(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        if self.storage == "fp16":
            t = t.clamp(-65504.0, 65504.0)
        t = t.to(self.tdtype).to(self.device)
        self._keep.append(t)
        return t.data_ptr()

    # ------------------------------------------------------------ weight forms
    def bn_fold(self, name):
        """BatchNormalization in inference mode as y = x*scale + shift."""
        p = self.w[name]
        eps = self.g.layers[name].config["epsilon"]
        inv = 1.0 / np.sqrt(p["moving_variance"].astype(np.float64) + eps)
        if "gamma" in p:
            inv = inv * p["gamma"].astype(np.float64)
        shift = p["beta"].astype(np.float64) - p["moving_mean"].astype(np.float64) * inv
        return inv.astype(np.float32), shift.astype(np.float32)

    def conv_weight(self, name, cin_pad=None):
        """HWIO -> [Cout][KH][KW][Cin(padded)]."""
        k = self.w[name]["kernel"]
        kh, kw, ci, co = k.shape
        cp = cin_pad or _ceil32(ci)
        out = np.zeros((co, kh, kw, cp), np.float32)
        out[..., :ci] = k.transpose(3, 0, 1, 2)
        return self.dev_h16(out)

    def convT_weight(self, name, cin_pad=None):
        """Keras (4,4,Cout,Cin) -> [phase a,b][Cout][dy][dx][Cin]; phase (a,b) computes
        out[2y+a, 2x+b] from in[y-1+a+dy, x-1+b+dx] with kernel tap (3-a-2dy, 3-b-2dx)."""
        k = self.w[name]["kernel"]
        _, _, co, ci = k.shape
        cp = cin_pad or _ceil32(ci)
        out = np.zeros((2, 2, co, 2, 2, cp), np.float32)
        for a in range(2):
            for b in range(2):
                for dy in range(2):
                    for dx in range(2):
                        out[a, b, :, dy, dx, :ci] = k[3 - a - 2 * dy, 3 - b - 2 * dx]
        return self.dev_h16(out)

    def stem_weight(self, name):
        """7x7x3 stem as a 7x1 convolution over 32 'channels' = 8 pixels x 4 padded
        channels of the prepared input; the 8th pixel and 4th channel get zero weight."""
        k = self.w[name]["kernel"]          # (7,7,3,64)
        out = np.zeros((64, 7, 1, 8, 4), np.float32)
        out[:, :, 0, :7, :3] = k.transpose(3, 0, 1, 2)
        return self.dev_h16(out.reshape(64, 7, 1, 32))

    # ------------------------------------------------------------------- ops
    def _add(self, rc, name, nbytes=0):
        L.check(rc)
        self.op_names.append(name)
        self.op_bytes.append(nbytes)

    def conv(self, name, x, wptr, cout, k, stride=1, pad=0, out=None, out_dtype=L.DT_BF16,
             scale=None, shift=None, act=L.ACT_NONE, residual=None, out2=None, scale2=None,
             shift2=None, in_scale=None, in_shift=None, transposed=False, collapse=None,
             temperature=1.0, tile_hint=0, build_only=False, res_after=False, res_shift=0, out2_linear=False):
        d = L.ConvDesc()
        cin = _ceil32(x.c)
        if x.c0 + cin > x.ctot:
            raise ValueError("%s: input slice needs %d channels of padding" % (name, cin - x.c))
        d.inp, d.weight = x.ptr, wptr
        d.N, d.H, d.W, d.Cin = x.n, x.h, x.w, cin
        if transposed:
            d.Ho, d.Wo, d.KH, d.KW = x.h, x.w, 2, 2
            d.stride_y = d.stride_x = 1
            d.pad_y = d.pad_x = 0
            oh, ow = 2 * x.h, 2 * x.w
        else:
            d.Ho = (x.h + 2 * pad - k) // stride + 1
            d.Wo = (x.w + 2 * pad - k) // stride + 1
            d.KH = d.KW = k
            d.stride_y = d.stride_x = stride
            d.pad_y = d.pad_x = pad
            oh, ow = d.Ho, d.Wo
        d.Cout = cout
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = x.pix, x.row, x.img
        for v, pre in ((out, "out"), (out2, "out2"), (residual, "res")):
            if v is not None:
                sh = res_shift if pre == "res" else 0
                assert (v.h, v.w) == (oh >> sh, ow >> sh) and v.c >= cout, (name, pre, v.h, v.w, v.c, oh, ow, cout)
                setattr(d, pre + "_pix_stride", v.pix)
                setattr(d, pre + "_row_stride", v.row)
                setattr(d, pre + "_img_stride", v.img)
        d.out = out.ptr if out is not None else None
        d.out2 = out2.ptr if out2 is not None else None
        d.residual = residual.ptr if residual is not None else None
        d.out_dtype = out_dtype
        d.res_dtype = L.DT_F32 if (residual is not None and residual.esize == 4) else L.DT_BF16
        d.scale = self.dev_f32(scale) if scale is not None else None
        d.shift = self.dev_f32(shift) if shift is not None else None
        d.scale2 = self.dev_f32(scale2) if scale2 is not None else None
        d.shift2 = self.dev_f32(shift2) if shift2 is not None else None
        if in_scale is not None:
            s = np.zeros(cin, np.float32)
            t = np.zeros(cin, np.float32)
            s[:x.c], t[:x.c] = in_scale, in_shift
            d.in_scale, d.in_shift = self.dev_f32(s), self.dev_f32(t)
        d.act = act
        d.transposed = 1 if transposed else 0
        d.temperature = temperature
        d.collapse_out = collapse.data_ptr() if collapse is not None else None
        d.collapse_value = self.collapse_value
        d.tile_hint = tile_hint
        d.fmt = self.fmt
        d.res_after_act = 1 if res_after else 0
        d.res_upsample_shift = res_shift
        d.out2_linear = 1 if out2_linear else 0
        pix_out = x.n * oh * ow
        nbytes = x.n * x.h * x.w * x.c * 2 + cout * d.KH * d.KW * cin * 2 * (4 if transposed else 1)
        if out is not None:
            nbytes += pix_out * cout * (4 if out_dtype == L.DT_F32 else 2)
        if out2 is not None:
            nbytes += pix_out * cout * 2
        if residual is not None:
            nbytes += pix_out * cout * residual.esize
        if collapse is not None:
            nbytes += pix_out * cout
        if build_only:
            return d, nbytes
        self._add(self.lib.miseg_plan_add_conv(self.plan, C.byref(d)), name, nbytes)

    def conv_bn_relu(self, base, x, k, out=None):
        """segmentation_models Conv2dBn: <base>_conv -> <base>_bn -> relu."""
        cout = self.w[base + "_conv"]["kernel"].shape[3]
        if out is None:
            out = self.buf((x.h), x.w, _ceil32(cout)).slice(0, cout)
        s, t = self.bn_fold(base + "_bn")
        # x may carry zero channels beyond the layer's own (padding to what a kernel takes)
        self.conv(base, x, self.conv_weight(base + "_conv", cin_pad=_ceil32(x.c)), cout, k, pad=(k - 1) // 2,
                  out=out, scale=s, shift=t, act=L.ACT_RELU)
        self.taps[base + "_relu"] = out
        return out

    def convT_bn_relu(self, tname, bnname, x, out=None, cin_pad=None):
        cout = self.w[tname]["kernel"].shape[2]
        if out is None:
            out = self.buf(2 * x.h, 2 * x.w, _ceil32(cout)).slice(0, cout)
        s, t = self.bn_fold(bnname)
        self.conv(tname, x, self.convT_weight(tname, cin_pad), cout, 4, out=out, scale=s, shift=t,
                  act=L.ACT_RELU, transposed=True)
        return out

    # ------------------------------------------------------------- encoders
    def _stem(self, conv_name, bn_name, in_bn, relu_out):
        h, w = self.h, self.wd
        padded = torch.zeros((self.n, h + 6, w + 8, 4), dtype=self.tdtype, device=self.device)
        self._keep.append(padded)
        if in_bn is not None:
            s, t = self.bn_fold(in_bn)
        else:
            s, t = np.ones(3, np.float32), np.zeros(3, np.float32)
        self._add(self.lib.miseg_plan_add_prep_input(
            self.plan, self.x_in.data_ptr(), padded.data_ptr(), self.dev_f32(s), self.dev_f32(t),
            self.n, h, w), "prep_input")
        d = L.ConvDesc()
        d.inp, d.weight = padded.data_ptr(), self.stem_weight(conv_name)
        d.N, d.H, d.W, d.Cin = self.n, h + 6, w // 2, 32
        d.Ho, d.Wo, d.Cout = h // 2, w // 2, 64
        d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = 7, 1, 2, 1, 0, 0
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = 8, (w + 8) * 4, (h + 6) * (w + 8) * 4
        d.out = relu_out.ptr
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = relu_out.pix, relu_out.row, relu_out.img
        s, t = self.bn_fold(bn_name)
        d.scale, d.shift = self.dev_f32(s), self.dev_f32(t)
        d.act, d.out_dtype = L.ACT_RELU, L.DT_BF16
        d.temperature = 1.0
        d.fmt = self.fmt
        self._add(self.lib.miseg_plan_add_conv(self.plan, C.byref(d)), conv_name)

    def _maxpool(self, x, out, scale=None, shift=None, name="pool"):
        self._add(self.lib.miseg_plan_add_maxpool3x3s2(
            self.plan, x.ptr, out.ptr, self.dev_f32(scale) if scale is not None else None,
            self.dev_f32(shift) if shift is not None else None, x.n, x.h, x.w, x.c, x.pix, out.pix),
            name)

    def _resnet(self, units, skip_dst, bottleneck=False):
        """skip_dst: {layer_name: View} destinations for the four skip tensors.  Pre-activation
        units: every BatchNorm + ReLU is the epilogue of the convolution that feeds it; the
        convolution that closes a unit adds the shortcut, writes the raw sum (the next unit's
        shortcut) and, as a second output, the next unit's bn1 + relu1 of it."""
        h2, w2 = self.h // 2, self.wd // 2
        relu0 = skip_dst.get("relu0") or self.buf(h2, w2, 64)
        self._stem("conv0", "bn0", "bn_data", relu0)
        self.taps["relu0"] = relu0
        # pooling0 with the first unit's bn1+relu1 applied to its output
        s, t = self.bn_fold("stage1_unit1_bn1")
        a = self.buf(h2 // 2, w2 // 2, 64)
        self._maxpool(relu0, a, s, t, "pooling0")
        self.taps["stage1_unit1_relu1"] = a
        order = [(s_, b_) for s_, reps in enumerate(units) for b_ in range(reps)]
        ex = 4 if bottleneck else 1
        xraw = None
        for idx, (stage, block) in enumerate(order):
            base = "stage%d_unit%d_" % (stage + 1, block + 1)
            f = 64 * 2 ** stage
            stride = 2 if (block == 0 and stage > 0) else 1
            ho, wo = a.h // stride, a.w // stride
            rdt, rdt_l = (self.tdtype, L.DT_BF16) if self.residual_bf16 else (torch.float32, L.DT_F32)
            if block == 0:
                shortcut = self.buf(ho, wo, f * ex, rdt)
                self.conv(base + "sc", a, self.conv_weight(base + "sc"), f * ex, 1, stride=stride,
                          out=shortcut, out_dtype=rdt_l)
            else:
                shortcut = xraw
            s, t = self.bn_fold(base + "bn2")
            if bottleneck:      # 1x1 -> 3x3 (carries the stride) -> 1x1 to four times the filters
                b1 = self.buf(a.h, a.w, f)
                self.conv(base + "conv1", a, self.conv_weight(base + "conv1"), f, 1, out=b1, scale=s, shift=t,
                          act=L.ACT_RELU)
                s, t = self.bn_fold(base + "bn3")
                b2 = self.buf(ho, wo, f)
                self.conv(base + "conv2", b1, self.conv_weight(base + "conv2"), f, 3, stride=stride, pad=1,
                          out=b2, scale=s, shift=t, act=L.ACT_RELU)
                last_name, last_in, last_k = base + "conv3", b2, 1
            else:
                b1 = self.buf(ho, wo, f)
                self.conv(base + "conv1", a, self.conv_weight(base + "conv1"), f, 3, stride=stride,
                          pad=1, out=b1, scale=s, shift=t, act=L.ACT_RELU)
                last_name, last_in, last_k = base + "conv2", b1, 3
            last = idx + 1 == len(order)
            if last:
                nbn, nname, need_raw = "bn1", "relu1", False
            else:
                ns, nb = order[idx + 1]
                nbn = "stage%d_unit%d_bn1" % (ns + 1, nb + 1)
                nname = "stage%d_unit%d_relu1" % (ns + 1, nb + 1)
                need_raw = nb != 0
            s2, t2 = self.bn_fold(nbn)
            a_next = skip_dst.get(nname) or self.buf(ho, wo, f * ex)
            xraw = self.buf(ho, wo, f * ex, rdt) if need_raw else None
            self.conv(last_name, last_in, self.conv_weight(last_name), f * ex, last_k, pad=(last_k - 1) // 2,
                      out=xraw, out_dtype=rdt_l, residual=shortcut, out2=a_next, scale2=s2,
                      shift2=t2)
            self.taps[nname] = a_next
            a = a_next
        return a

    def _densenet(self, block_layers, skip_dst):
        h2, w2 = self.h // 2, self.wd // 2
        relu = skip_dst.get("conv1/relu") or self.buf(h2, w2, 64)
        self._stem("conv1/conv", "conv1/bn", None, relu)
        self.taps["conv1/relu"] = relu
        c = 64
        h, w = h2 // 2, w2 // 2
        cat = self.buf(h, w, c + 32 * block_layers[0])
        self._maxpool(relu, cat.slice(0, c), name="pool1")
        for bi, nl in enumerate(block_layers):
            blk = "conv%d" % (bi + 2)
            mid = self.buf(h, w, 128) if not self.fuse_dense else None
            nomid = View(torch.empty((0, 128), dtype=self.tdtype, device=self.device), self.n, h, w, 128)
            for li in range(nl):
                n = "%s_block%d" % (blk, li + 1)
                s0, t0 = self.bn_fold(n + "_0_bn")
                s1, t1 = self.bn_fold(n + "_1_bn")
                fused = False
                if self.fuse_dense:
                    d1, n1 = self.conv(n + "_1_conv", cat.slice(0, c), self.conv_weight(n + "_1_conv"), 128, 1,
                                       scale=s1, shift=t1, act=L.ACT_RELU, in_scale=s0, in_shift=t0,
                                       build_only=True)
                    d2, n2 = self.conv(n + "_2_conv", nomid, self.conv_weight(n + "_2_conv"), 32, 3, pad=1,
                                       out=cat.slice(c, 32), build_only=True)
                    d2.inp = None
                    n2 -= self.n * h * w * 128 * 2       # the 3x3's input never reaches HBM
                    rc = self.lib.miseg_plan_add_dense_layer(self.plan, C.byref(d1), C.byref(d2))
                    # the one-kernel layer keeps the input BatchNorm's coefficients in LDS and takes as many
                    # input channels as fit there (DenseNet-201's last conv5 layers exceed it on bf16
                    # storage): such a layer runs as its two convolutions
                    fused = rc != L.E_UNSUPPORTED_SHAPE
                    if fused:
                        self._add(rc, n + "_1_conv+_2_conv", n1 + n2)
                if not fused:
                    if mid is None:
                        mid = self.buf(h, w, 128)
                    self.conv(n + "_1_conv", cat.slice(0, c), self.conv_weight(n + "_1_conv"), 128, 1,
                              out=mid, scale=s1, shift=t1, act=L.ACT_RELU, in_scale=s0, in_shift=t0)
                    self.conv(n + "_2_conv", mid, self.conv_weight(n + "_2_conv"), 32, 3, pad=1,
                              out=cat.slice(c, 32))
                c += 32
            self.taps[blk + "_out"] = cat.slice(0, c)
            if bi + 1 < len(block_layers):
                tname = "pool%d" % (bi + 2)
                s0, t0 = self.bn_fold(tname + "_bn")
                tconv = skip_dst.get(tname + "_conv") or self.buf(h, w, c // 2)
                self.conv(tname + "_conv", cat.slice(0, c), self.conv_weight(tname + "_conv"),
                          c // 2, 1, out=tconv, in_scale=s0, in_shift=t0)
                self.taps[tname + "_conv"] = tconv
                c //= 2
                h, w = h // 2, w // 2
                cat = self.buf(h, w, c + 32 * block_layers[bi + 1])
                self._add(self.lib.miseg_plan_add_avgpool2x2(
                    self.plan, tconv.ptr, cat.ptr, self.n, tconv.h, tconv.w, c, tconv.pix, cat.pix),
                    tname + "_pool")
        s, t = self.bn_fold("bn")
        out = self.buf(h, w, c)
        self._add(self.lib.miseg_plan_add_scale_shift_relu(
            self.plan, cat.ptr, out.ptr, self.dev_f32(s), self.dev_f32(t), self.n * h * w, c,
            cat.pix, out.pix), "bn+relu")
        self.taps["relu"] = out
        return out

    def _vgg(self, blocks, skip_dst):
        """keras_applications VGG-16/19: 3x3 convolutions with bias and ReLU (the epilogue's shift and
        activation), MaxPooling2D(2) after each block.  block1_conv1 reads the fp32 image through its
        3x3 neighbourhoods laid out as 32 channels, i.e. as a 1x1 convolution."""
        col = self.buf(self.h, self.wd, 32)
        self._add(self.lib.miseg_plan_add_im2col3x3(self.plan, self.x_in.data_ptr(), col.ptr, self.n,
                                                    self.h, self.wd), "im2col3x3")
        x = col
        for b, reps in enumerate(blocks):
            for i in range(reps):
                name = "block%d_conv%d" % (b + 1, i + 1)
                k = self.w[name]["kernel"]
                cout = k.shape[3]
                out = (skip_dst.get(name) if i + 1 == reps else None) or self.buf(x.h, x.w, cout)
                if x is col:
                    w1 = np.zeros((cout, 1, 1, 32), np.float32)
                    w1[:, 0, 0, :27] = k.transpose(3, 0, 1, 2).reshape(cout, 27)
                    self.conv(name, x, self.dev_h16(w1), cout, 1, out=out, shift=self.w[name]["bias"],
                              act=L.ACT_RELU)
                else:
                    self.conv(name, x, self.conv_weight(name), cout, 3, pad=1, out=out,
                              shift=self.w[name]["bias"], act=L.ACT_RELU)
                self.taps[name] = out
                x = out
            pooled = self.buf(x.h // 2, x.w // 2, x.c)
            self._add(self.lib.miseg_plan_add_maxpool2x2(self.plan, x.ptr, pooled.ptr, self.n, x.h, x.w,
                                                         x.c, x.pix, pooled.pix), "block%d_pool" % (b + 1))
            x = pooled
        return x

    # -------------------------------------------------------------- decoders
    def _head(self, name, x, build_only=False):
        act = L.ACT_SIGMOID if self.g.activation == "sigmoid" else L.ACT_SOFTMAX
        temp = self.g.temperature if self.g.temperature is not None else 1.0
        out = View(self.probs, self.n, self.h, self.wd, self.classes)
        return self.conv(name, x, self.conv_weight(name), self.classes, 3, pad=1, out=out,
                         out_dtype=L.DT_F32, shift=self.w[name]["bias"], act=act, collapse=self.mask,
                         temperature=temp, build_only=build_only)

    def _unet_tail(self, base, x, head):
        """decoder_stage4b (Conv2dBn 3x3 32 -> 32) + final_conv + activation as one kernel: the
        32-channel full-resolution tensor between the two is not written to HBM."""
        cout = self.w[base + "_conv"]["kernel"].shape[3]
        s, t = self.bn_fold(base + "_bn")
        body, nb = self.conv(base, x, self.conv_weight(base + "_conv"), cout, 3, pad=1, scale=s, shift=t,
                             act=L.ACT_RELU, build_only=True)
        # descriptor of the head as it would run on the (virtual) 32-channel tensor
        virt = View(torch.empty((0, 32), dtype=self.tdtype, device=self.device), self.n, x.h, x.w, cout)
        hd, nh = self._head(head, virt, build_only=True)
        hd.inp = None
        nh -= self.n * x.h * x.w * cout * 2
        self._add(self.lib.miseg_plan_add_conv_c32_head(self.plan, C.byref(body), C.byref(hd)),
                  base + "+" + head, nb + nh)

    def _linknet_tail_ok(self, x):
        w = self.w
        return (x.c == 64 and self.classes <= 4 and
                w["decoder_stage4a_conv"]["kernel"].shape[3] == 16 and
                w["decoder_stage4b_transpose"]["kernel"].shape[2:] == (16, 16) and
                w["decoder_stage4c_conv"]["kernel"].shape[2:] == (16, 16) and
                w["final_conv"]["kernel"].shape[:3] == (3, 3, 16))

    def _linknet_tail(self, x):
        """decoder_stage4 (1x1 64 -> 16, transpose 16 -> 16, 1x1 16 -> 16, each BN + ReLU) + final_conv +
        activation + collapse as one kernel: none of the 16-channel full-resolution tensors reaches HBM."""
        d = L.LinknetTailDesc()
        d.inp = x.ptr
        d.wa = self.conv_weight("decoder_stage4a_conv", cin_pad=64)
        d.wb = self.convT_weight("decoder_stage4b_transpose", cin_pad=16)
        d.wc = self.conv_weight("decoder_stage4c_conv", cin_pad=32)
        k = self.w["final_conv"]["kernel"]                      # (3, 3, 16, classes)
        wd = np.zeros((16, 160), np.float32)
        wd[:self.classes, :144] = k.transpose(3, 0, 1, 2).reshape(self.classes, 144)
        d.wd = self.dev_h16(wd)
        for key, bn in (("a", "decoder_stage4a_bn"), ("b", "decoder_stage4b_bn"), ("c", "decoder_stage4c_bn")):
            sc, sh = self.bn_fold(bn)
            setattr(d, "s" + key, self.dev_f32(sc))
            setattr(d, "t" + key, self.dev_f32(sh))
        d.bias = self.dev_f32(self.w["final_conv"]["bias"])
        d.out, d.collapse_out = self.probs.data_ptr(), self.mask.data_ptr()
        d.N, d.H, d.W = x.n, x.h, x.w
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = x.pix, x.row, x.img
        c = self.classes
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = c, c * self.wd, c * self.wd * self.h
        d.classes = c
        d.act = L.ACT_SIGMOID if self.g.activation == "sigmoid" else L.ACT_SOFTMAX
        d.temperature = self.g.temperature if self.g.temperature is not None else 1.0
        d.collapse_value = self.collapse_value
        d.fmt = self.fmt
        nbytes = x.n * x.h * x.w * 64 * 2 + self.n * self.h * self.wd * c * 5
        self._add(self.lib.miseg_plan_add_linknet_tail(self.plan, C.byref(d)), "decoder_stage4+final_conv", nbytes)

    def _linknet_block_ok(self, i, x, skip):
        w = self.w
        pre = "decoder_stage%d" % i
        m = x.c // 4
        return (x.c in (64, 128) and skip is not None and skip.c == 64 and
                w[pre + "a_conv"]["kernel"].shape[3] == m and
                w[pre + "b_transpose"]["kernel"].shape[2:] == (m, m) and
                w[pre + "c_conv"]["kernel"].shape[2:] == (m, 64))

    def _linknet_block(self, i, x, skip):
        """A decoder block with C = 64 or 128 input channels and a 64-channel skip tensor (decoder_stage3: 1x1
        C -> C/4, transpose C/4 -> C/4, 1x1 C/4 -> 64, each BN + ReLU, + skip) as one kernel: neither C/4-channel
        tensor reaches HBM.  Returns the block's output."""
        pre = "decoder_stage%d" % i
        out = self.buf(2 * x.h, 2 * x.w, 64)
        d = L.LinknetBlockDesc()
        d.inp = x.ptr
        d.wa = self.conv_weight(pre + "a_conv", cin_pad=x.c)
        d.wb = self.convT_weight(pre + "b_transpose", cin_pad=x.c // 4)
        d.wc = self.conv_weight(pre + "c_conv", cin_pad=32)
        d.cin = x.c
        for key, bn in (("a", pre + "a_bn"), ("b", pre + "b_bn"), ("c", pre + "c_bn")):
            sc, sh = self.bn_fold(bn)
            setattr(d, "s" + key, self.dev_f32(sc))
            setattr(d, "t" + key, self.dev_f32(sh))
        d.skip, d.out = skip.ptr, out.ptr
        d.N, d.H, d.W = x.n, x.h, x.w
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = x.pix, x.row, x.img
        d.skip_pix_stride, d.skip_row_stride, d.skip_img_stride = skip.pix, skip.row, skip.img
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = out.pix, out.row, out.img
        d.fmt = self.fmt
        nbytes = x.n * x.h * x.w * x.c * 2 + 2 * x.n * out.h * out.w * 64 * 2
        self._add(self.lib.miseg_plan_add_linknet_block(self.plan, C.byref(d)), pre, nbytes)
        return out

    def _fpn_head_gather(self, x):
        """final_upsampling (bilinear x2) + head_conv + activation + collapse with the channel contraction at
        HALF resolution: both layers are linear, so the nine taps' contractions W[tap][:, c] . x run as one 1x1
        convolution with 9 * classes fp32 output channels where x lives, and miseg_head_gather_fwd samples,
        sums and activates per output pixel (the 3x3 zero padding = taps outside the image skipped)."""
        c = self.classes
        k = self.w["head_conv"]["kernel"]                       # (3, 3, Cin, classes)
        cy = _ceil32(9 * c)
        wy = np.zeros((cy, 1, 1, _ceil32(x.c)), np.float32)
        for tap in range(9):
            wy[tap * c:tap * c + c, 0, 0, :k.shape[2]] = k[tap // 3, tap % 3].T
        ybuf = torch.zeros((self.n, x.h, x.w, cy), dtype=torch.float32, device=self.device)
        self._keep.append(ybuf)
        self.conv("head_conv_taps", x, self.dev_h16(wy), cy, 1, out=View(ybuf, self.n, x.h, x.w, cy),
                  out_dtype=L.DT_F32)
        act = L.ACT_SIGMOID if self.g.activation == "sigmoid" else L.ACT_SOFTMAX
        temp = self.g.temperature if self.g.temperature is not None else 1.0
        self._add(self.lib.miseg_plan_add_head_gather(
            self.plan, ybuf.data_ptr(), self.dev_f32(self.w["head_conv"]["bias"]), self.probs.data_ptr(),
            self.mask.data_ptr(), self.n, x.h, x.w, c, cy, act, temp, self.collapse_value),
            "final_upsampling+head_conv", self.n * x.h * x.w * cy * 4 + self.n * self.h * self.wd * c * 5)

    def _fpn_head(self, x):
        """final_upsampling (bilinear x2) + head_conv + activation + collapse as one kernel: the
        up-sampled 128-channel tensor (256 bytes per output pixel) never reaches HBM."""
        d = L.FpnHeadDesc()
        d.inp = x.ptr
        k = self.w["head_conv"]["kernel"]                       # (3, 3, 128, classes)
        wd = np.zeros((16, 9, 128), np.float32)
        wd[:self.classes] = k.transpose(3, 0, 1, 2).reshape(self.classes, 9, 128)
        d.wd = self.dev_h16(wd)
        d.bias = self.dev_f32(self.w["head_conv"]["bias"])
        d.out, d.collapse_out = self.probs.data_ptr(), self.mask.data_ptr()
        d.N, d.H, d.W = x.n, x.h, x.w
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = x.pix, x.row, x.img
        c = self.classes
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = c, c * self.wd, c * self.wd * self.h
        d.classes = c
        d.act = L.ACT_SIGMOID if self.g.activation == "sigmoid" else L.ACT_SOFTMAX
        d.temperature = self.g.temperature if self.g.temperature is not None else 1.0
        d.collapse_value = self.collapse_value
        d.fmt = self.fmt
        nbytes = x.n * x.h * x.w * 128 * 2 + self.n * self.h * self.wd * c * 5
        self._add(self.lib.miseg_plan_add_fpn_head(self.plan, C.byref(d)), "final_upsampling+head_conv", nbytes)

    @staticmethod
    def _t2_weights(k):
        """Phase weights [a, b][Cout][dy][dx][Cin] of `3x3 convolution (HWIO kernel k) after nearest x2
        up-sampling` in the layout of the transpose-convolution kernels: the 3x3 taps that fall on the same
        source pixel are summed (a = 0: rows y-1 <- ky 0, y <- ky 1 + ky 2; a = 1: y <- ky 0 + ky 1,
        y+1 <- ky 2; columns alike)."""
        rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
        ci, co = k.shape[2], k.shape[3]
        wt = np.zeros((2, 2, co, 2, 2, ci), np.float32)
        for a in range(2):
            for b in range(2):
                for dy in range(2):
                    for dx in range(2):
                        acc = np.zeros((ci, co), np.float32)
                        for ky in rows[a][dy]:
                            for kx in rows[b][dx]:
                                acc += k[ky, kx]
                        wt[a, b, :, dy, dx, :] = acc.T
        return wt

    def _fpn_segm2a_split(self, q3, skip):
        """segm_stage2a on p2 = fpn_stage_p2_conv(skip) + bias + up2(p3) without p2: by linearity (zero padding
        included) conv3x3(p2) = conv3x3(W_l skip) + conv3x3(up2(p3 + bias)), i.e. a 3x3 convolution of the
        64-channel skip tensor with the composed kernel W_l W3[ky, kx] plus the phase convolution (see
        _fpn_final_stage_split) of q3 = p3 + bias, which the p3 layer writes as its second output.  The
        256-channel full-resolution tensor p2 and its 1x1 convolution never run."""
        k3 = self.w["segm_stage2a_conv"]["kernel"].astype(np.float32)       # (3, 3, 256, 128)
        kl = self.w["fpn_stage_p2_conv"]["kernel"].astype(np.float32)[0, 0]  # (Cskip, 256)
        sc, sh = self.bn_fold("segm_stage2a_bn")
        part = self.buf(2 * q3.h, 2 * q3.w, 128)
        self.conv("segm_stage2a_coarse", q3, self.dev_h16(self._t2_weights(k3)), 128, 4, out=part, scale=sc,
                  transposed=True)
        cs = kl.shape[0]
        wc = np.zeros((128, 3, 3, _ceil32(cs)), np.float32)
        for ky in range(3):
            for kx in range(3):
                wc[:, ky, kx, :cs] = (kl.astype(np.float64) @ k3[ky, kx].astype(np.float64)).T
        out = self.buf(skip.h, skip.w, 128)
        self.conv("segm_stage2a", skip, self.dev_h16(wc), 128, 3, pad=1, out=out, scale=sc, shift=sh,
                  act=L.ACT_RELU, residual=part)
        self.taps["segm_stage2a_relu"] = out
        return out

    def _fpn_final_stage_split(self, pyr):
        """final_stage (3x3, 512 -> 128, on [s2, up2(s3), up4(s4), up8(s5)]) without the up-sampled tensors.
        A 3x3 convolution of a nearest-x2 up-sampled tensor is, per output phase (row parity a, column
        parity b), a 2x2 convolution of the tensor itself whose taps are sums of the 3x3 taps that fall on
        the same source pixel (a = 0: rows y-1 <- ky 0, y <- ky 1 + ky 2; a = 1: y <- ky 0 + ky 1,
        y+1 <- ky 2): exactly the four phase convolutions of the transpose-convolution kernels.  So
            final_stage = BN(conv3x3(s2; W[:, :, :128]) + T2([s3, up2(s4), up4(s5)]; W[:, :, 128:]))
        with the three coarser tensors gathered at HALF resolution: 58 % of the layer's multiply-adds, a
        quarter of the up-sampling traffic, and the 512-channel full-resolution buffer is gone.  The
        BatchNorm scale goes onto the T2 part in its own epilogue, which leaves (scale * conv + shift)
        + residual, then ReLU, for the epilogue of the 3x3 part."""
        h2, w2 = 2 * pyr[2].h, 2 * pyr[2].w
        s2 = self.buf(h2, w2, 128)
        half = self.buf(h2 // 2, w2 // 2, 384)            # [s3, up2(s4), up4(s5)]
        for p, st, shift in zip(pyr, (5, 4, 3, 2), (3, 2, 1, 0)):
            s = self._fpn_segm2a_split(*p) if isinstance(p, tuple) else self.conv_bn_relu("segm_stage%da" % st, p, 3)
            if shift == 0:
                self.conv_bn_relu("segm_stage%db" % st, s, 3, out=s2)
            elif shift == 1:
                self.conv_bn_relu("segm_stage%db" % st, s, 3, out=half.slice(0, 128))
            else:
                s = self.conv_bn_relu("segm_stage%db" % st, s, 3)
                dst = half.slice(128 * (shift - 1), 128)
                self._add(self.lib.miseg_plan_add_upsample_nearest(
                    self.plan, s.ptr, None, dst.ptr, self.n, s.h, s.w, 128, shift - 1, s.pix, 0, dst.pix),
                    "upsampling_stage%d" % st)
        k = self.w["final_stage_conv"]["kernel"].astype(np.float32)       # (3, 3, 512, 128)
        sc, sh = self.bn_fold("final_stage_bn")
        wt = self._t2_weights(k[:, :, 128:, :])
        part = self.buf(h2, w2, 128)
        self.conv("final_stage_coarse", half, self.dev_h16(wt), 128, 4, out=part, scale=sc, transposed=True)
        fine = np.zeros((128, 3, 3, 128), np.float32)
        fine[:] = k[:, :, :128, :].transpose(3, 0, 1, 2)
        out = self.buf(h2, w2, 128)
        self.conv("final_stage", s2, self.dev_h16(fine), 128, 3, pad=1, out=out, scale=sc, shift=sh,
                  act=L.ACT_RELU, residual=part)
        self.taps["final_stage_relu"] = out
        return out

    def _lower(self):
        g = self.g
        enc, dec = g.encoder, g.decoder
        skip_names = list(g.skip_layers)
        skip_syms = [g.layers[n].out for n in skip_names]
        enc_out = g.layers[g.encoder_output].out
        skip_dst = {}
        concat = []
        if dec == "U-Net":
            # decoder_stage{i}_concat = [transpose-conv output (f channels), skip]
            h, w = enc_out.h, enc_out.w
            for i, f in enumerate((512, 256, 128, 64, 32)):
                h, w = 2 * h, 2 * w
                sc = skip_syms[i].c if i < len(skip_syms) else 0
                # more than 32 channels: padded (zero channels, zero weights) to a multiple of 64, which is
                # what the LDS-DMA kernels take
                cb = self.buf(h, w, _ceil32(f + sc) if f + sc <= 32 else (f + sc + 63) // 64 * 64)
                concat.append(cb)
                if sc:
                    skip_dst[skip_names[i]] = cb.slice(f, sc)
        if enc in RESNET_UNITS:
            x = self._resnet(RESNET_UNITS[enc], skip_dst, enc in RESNET_BOTTLENECK)
        elif enc in VGG_BLOCKS:
            x = self._vgg(VGG_BLOCKS[enc], skip_dst)
        else:
            x = self._densenet(DENSENET_LAYERS[enc], skip_dst)
        skips = [self.taps[n] for n in skip_names]
        if dec != "FPN" and "center_block1_conv" in self.w:
            # the backbone ends in a pooling layer (the VGGs): two Conv3x3BnReLU(512) blocks first
            x = self.conv_bn_relu("center_block1", x, 3)
            x = self.conv_bn_relu("center_block2", x, 3)
        if dec == "U-Net":
            for i, f in enumerate((512, 256, 128, 64, 32)):
                cb = concat[i]
                self.convT_bn_relu("decoder_stage%da_transpose" % i, "decoder_stage%da_bn" % i, x,
                                   out=cb.slice(0, f))
                xin = cb.slice(0, f + (skips[i].c if i < len(skips) else 0))
                if i == 4 and self.fuse_tail and _ceil32(xin.c) == 32 and self.classes <= 4 and \
                        self.w["decoder_stage4b_conv"]["kernel"].shape[3] == 32:
                    self._unet_tail("decoder_stage4b", xin, "final_conv")
                    x = None
                else:
                    x = self.conv_bn_relu("decoder_stage%db" % i, cb, 3)
            if x is not None:
                self._head("final_conv", x)
        elif dec == "Linknet":
            for i in range(5):
                if i == 4 and self.fuse_tail and self._linknet_tail_ok(x):
                    self._linknet_tail(x)
                    x = None
                    break
                if self.fuse_tail and i < len(skips) and self._linknet_block_ok(i, x, skips[i]):
                    x = self._linknet_block(i, x, skips[i])
                    self.taps["decoder_stage%d_out" % i] = x
                    continue
                # the transpose convolution's input channels (C / 4) padded to a multiple of 64 with zero
                # channels and zero weights: the ring kernel takes such a layer, the generic one is 1.5-2x slower
                ca = self.w["decoder_stage%da_conv" % i]["kernel"].shape[3]
                cpad = (ca + 63) // 64 * 64
                xa = self.buf(x.h, x.w, cpad)
                self.conv_bn_relu("decoder_stage%da" % i, x, 1, out=xa.slice(0, ca))
                xb = self.buf(2 * x.h, 2 * x.w, cpad)       # the same padding for the 1x1 convolution behind it
                self.convT_bn_relu("decoder_stage%db_transpose" % i, "decoder_stage%db_bn" % i, xa,
                                   out=xb.slice(0, ca), cin_pad=cpad)
                x = xb
                base = "decoder_stage%dc" % i
                cout = self.w[base + "_conv"]["kernel"].shape[3]
                out = self.buf(x.h, x.w, _ceil32(cout)).slice(0, cout)
                s, t = self.bn_fold(base + "_bn")
                # Add([relu(bn(conv)), skip]): the skip is added after the activation, in the
                # convolution's epilogue
                self.conv(base, x, self.conv_weight(base + "_conv", cin_pad=cpad), cout, 1, out=out, scale=s,
                          shift=t, act=L.ACT_RELU, residual=skips[i] if i < len(skips) else None,
                          res_after=i < len(skips))
                self.taps["decoder_stage%d_out" % i] = out
                x = out
            if x is not None:
                self._head("final_conv", x)
        else:  # FPN
            pyr = []
            for j, st in enumerate((5, 4, 3, 2)):
                pre = "fpn_stage_p%d_pre_conv" % st
                if pre in self.w:
                    y = self.buf(x.h, x.w, 256)
                    self.conv(pre, x, self.conv_weight(pre), 256, 1, out=y, shift=self.w[pre]["bias"])
                    x = y
                lat = "fpn_stage_p%d_conv" % st
                sk = skips[j]
                if st == 2 and self.split_fpn_stage:
                    # the finest level is consumed by segm_stage2a only: folded into it (_fpn_segm2a_split)
                    pyr.append((q3, sk))
                    continue
                p = self.buf(sk.h, sk.w, 256)
                q3 = self.buf(sk.h, sk.w, 256) if st == 3 and self.split_fpn_stage else None
                # p = lateral(skip) + nearest_upsample_x2(x): the add in the convolution's epilogue; the
                # level above the finest one also leaves p + (bias of the finest level's lateral convolution)
                self.conv(lat, sk, self.conv_weight(lat), 256, 1, out=p, shift=self.w[lat]["bias"],
                          residual=x, res_shift=1, out2=q3,
                          scale2=np.ones(256, np.float32) if q3 is not None else None,
                          shift2=self.w["fpn_stage_p2_conv"]["bias"] if q3 is not None else None,
                          out2_linear=q3 is not None)
                self.taps["fpn_stage_p%d_add" % st] = p
                pyr.append(p)
                x = p
            if self.split_fpn_stage:
                x = self._fpn_final_stage_split(pyr)
            else:
                p2 = pyr[3]
                agg = self.buf(p2.h, p2.w, 512)        # [s2, s3, s4, s5]
                for p, st, slot, shift in zip(pyr, (5, 4, 3, 2), (3, 2, 1, 0), (3, 2, 1, 0)):
                    s = self.conv_bn_relu("segm_stage%da" % st, p, 3)
                    if shift == 0:
                        self.conv_bn_relu("segm_stage%db" % st, s, 3, out=agg.slice(128 * slot, 128))
                    else:
                        s = self.conv_bn_relu("segm_stage%db" % st, s, 3)
                        dst = agg.slice(128 * slot, 128)
                        self._add(self.lib.miseg_plan_add_upsample_nearest(
                            self.plan, s.ptr, None, dst.ptr, self.n, s.h, s.w, 128, shift, s.pix, 0,
                            dst.pix), "upsampling_stage%d" % st)
                x = self.conv_bn_relu("final_stage", agg, 3)
            if self.fuse_tail and self.head_gather and self.classes <= 4:
                self._fpn_head_gather(x)
            elif self.fuse_tail and x.c == 128 and self.classes <= 4 and \
                    self.w["head_conv"]["kernel"].shape[:3] == (3, 3, 128):
                self._fpn_head(x)
            else:
                up = self.buf(2 * x.h, 2 * x.w, 128)
                self._add(self.lib.miseg_plan_add_upsample_bilinear2x(
                    self.plan, x.ptr, up.ptr, self.n, x.h, x.w, 128, x.pix, up.pix), "final_upsampling")
                self._head("head_conv", up)
        self.num_ops = self.lib.miseg_plan_num_ops(self.plan)

    # ---------------------------------------------------------------- running
    def stream_ptr(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def run(self):
        """Launch the whole plan on torch's current stream (asynchronous)."""
        L.check(self.lib.miseg_plan_run(self.plan, self.stream_ptr()))

    def forward(self, x):
        """x: (n,H,W,3) float32 tensor or array with n <= batch.  Returns the fp32
        probability tensor (n,H,W,classes) on the device (a view of a static buffer)."""
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        n = x.shape[0]
        if tuple(x.shape[1:]) != (self.h, self.wd, 3) or n > self.n:
            raise ValueError("expected (<=%d, %d, %d, 3), got %r" % (self.n, self.h, self.wd, tuple(x.shape)))
        self.x_in[:n].copy_(x.to(torch.float32), non_blocking=True)
        self.run()
        return self.probs[:n]

    def time_ops(self, first, last):
        ms = C.c_float(0)
        L.check(self.lib.miseg_plan_run_range(self.plan, first, last, self.stream_ptr(), C.byref(ms)))
        return ms.value

    def op_macs(self, i):
        return int(self.lib.miseg_plan_op_macs(self.plan, i))

    def op_kernel(self, i):
        buf = C.create_string_buffer(128)
        L.check(self.lib.miseg_plan_op_kernel(self.plan, i, buf, 128))
        return buf.value.decode()
