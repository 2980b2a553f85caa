"""Training step (forward with batch statistics, backward, Keras-form Adam) for every
network `make_model` builds, executed layer by layer over the netspec.Graph on libmiseg.so
kernels: the replacement of the tf.keras `Model.fit` inner loop the reference drives at
/root/reference model_train.py:284-329 (loss and metrics chosen at :172-179, Adam at
utils/model.py:234).

Layout and precision: activations and activation gradients are bf16 NHWC (channels padded to a
multiple of 32 with zeros), accumulation is fp32, parameters live in three flat fp32 buffers
(value, Adam m, Adam v) plus a flat fp32 gradient buffer and a flat bf16 working copy that the
forward convolutions read directly.  Convolution kernels are stored in the layout the kernels
consume, [Cout][KH][KW][Cin] (transpose-conv: [phase][Cout][2][2][Cin]), so the weight-gradient
kernel writes straight into the gradient buffer and Adam is one launch over the flat buffers.
Data gradients reuse the forward convolution kernels on re-packed (flipped / transposed)
weights.  torch supplies device memory, streams, the tiny weight re-packing copies and the
collective (RCCL all-reduce of the flat gradient buffer in buckets, issued while the backward
pass is still running).
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import lib as L


STAT_REPLICAS = 32


def _c32(c):
    return (c + 31) // 32 * 32


class Act:
    """bf16 NHWC activation (or gradient) view: channels [c0, c0 + c) of `t`.
    pro = (scale, shift): the activation is relu(t * scale + shift) of the stored tensor -- a BatchNormalization +
    ReLU whose apply pass did not run because the one layer that reads the result applies it on the way in
    (miseg_conv_desc.in_scale, miseg_wgrad_desc.x_scale)."""

    def __init__(self, t, c, c0=0, pro=None):
        self.t, self.c, self.c0, self.pro = t, c, c0, pro

    @property
    def n(self):
        return self.t.shape[0]

    @property
    def h(self):
        return self.t.shape[1]

    @property
    def w(self):
        return self.t.shape[2]

    @property
    def pix(self):
        return self.t.shape[3]

    @property
    def ptr(self):
        return self.t.data_ptr() + 2 * self.c0

    @property
    def pixels(self):
        return self.n * self.h * self.w

    def slice(self, c0, c):
        assert self.pro is None, "a tensor with a pending BatchNorm + ReLU has one reader, which takes all of it"
        return Act(self.t, c, self.c0 + c0)

    def float(self):
        v = self.t[..., self.c0:self.c0 + self.c].float()
        if self.pro is not None:
            v = torch.relu(v * self.pro[0][:self.c] + self.pro[1][:self.c]).to(torch.bfloat16).float()
        return v


class Seg:
    """One parameter tensor inside the flat buffers."""

    def __init__(self, off, count, shape):
        self.off, self.count, self.shape = off, count, shape


class Trainer:
    def __init__(self, graph, weights, batch, device="cuda:0", learning_rate=1e-3, beta_1=0.9, side_stream=True,
                 beta_2=0.999, epsilon=1e-7, loss="cce_dice", world_size=1, bucket_mb=25,
                 freeze_encoder=False):
        if not torch.cuda.is_available():
            raise L.MisegError("no GPU visible: the HIP training path has no CPU fallback")
        from .profile import TimedLib
        self.lib = TimedLib(L.load())      # transparent unless profile_step() switches it on
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        # Weight gradients are off the critical path of the backward pass (nothing downstream reads
        # them before the optimiser / the all-reduce): they run on a second stream, next to the
        # data-gradient chain, and fill the ramps and tails of its launches.
        self.side = torch.cuda.Stream(self.dev) if side_stream else None
        L.check(self.lib.miseg_device_check())
        self.g = graph
        self.n = batch
        self.H, self.W, _ = graph.input_shape
        self.classes = graph.classes
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.step_count = 0
        from . import losses as LS
        if loss not in LS.LOSS_FLAGS:
            raise NotImplementedError("loss %r is not built on the HIP path" % (loss,))
        self.loss_flags = LS.LOSS_FLAGS[loss]
        self.focal = bool(self.loss_flags & L.LOSS_FOCAL)
        self.focal_alpha, self.focal_gamma = LS.FOCAL_ALPHA, LS.FOCAL_GAMMA
        self.world = world_size
        # encoder_freeze (utils/model.py:179,191,203): the backbone's layers keep their weights and
        # their BatchNormalization layers run on the moving statistics
        self.frozen = set()
        if freeze_encoder:
            for name in graph.layers:
                self.frozen.add(name)
                if name == graph.encoder_output:
                    break
        self.bucket_bytes = int(bucket_mb * 2 ** 20)
        self._analyse()
        self._alloc_params(weights)
        self.keep_grads = False       # debugging: keep (dy, dx) of every layer of the last backward
        self.grad_log = None

    # ------------------------------------------------------------------ graph analysis
    def _analyse(self):
        g = self.g
        self.consumers = {n: [] for n in g.layers}
        for l in g.layers.values():
            for i in l.inbound:
                self.consumers[i].append(l.name)
        self.order = list(g.layers)
        # Concatenate without copies: a tensor that feeds a Concatenate is produced directly inside
        # the concatenated buffer (every kernel writes through a pixel stride), nested concatenations
        # (DenseNet blocks) share the outermost buffer.  place[tensor] = (concat layer, channel offset)
        self.place = {}
        for l in g.layers.values():
            if l.cls != "Concatenate":
                continue
            chans = [g.layers[i].out.c for i in l.inbound]
            if any(c % 32 for c in chans):
                continue
            c0 = 0
            for i, c in zip(l.inbound, chans):
                if i not in self.place and g.layers[i].cls not in ("InputLayer", "ZeroPadding2D"):
                    self.place[i] = (l.name, c0)
                c0 += c
        self.head_up = self._fused_head_source()
        self.fs = self._fused_final_stage()
        self.p2 = self._folded_finest_level()
        self._ones = {}
        if self.fs is not None:
            # the aggregated buffer does not exist: its first input gets a tensor of its own
            for i in list(self.place):
                if self.place[i][0] == self.fs["cat"]:
                    del self.place[i]

    def _fused_head_source(self):
        """Name of the UpSampling2D(2, 'bilinear') whose only consumer is the 3x3 head convolution (the FPN
        decoder's final_upsampling + head_conv), else None.  Both layers are linear, so the head's nine
        channel contractions run at half resolution as one 1x1 convolution with 9 * classes outputs and a
        gather kernel samples and sums them; the backward is the adjoint gather followed by the 1x1
        convolution's gradients.  Neither the up-sampled tensor nor its gradient exists."""
        g = self.g
        head = g.layers[g.logits_layer]
        if os.environ.get("MISEG_NO_FUSED_HEAD") or head.cls != "Conv2D" or head.name in self.frozen:
            return None
        cfg = head.config
        if cfg["kernel_size"] != 3 or cfg["strides"] != 1 or cfg["padding"] != "same" or head.out.c > 3 or \
                cfg.get("activation") == "relu":
            return None
        up = g.layers[head.inbound[0]]
        if up.cls != "UpSampling2D" or up.config["interpolation"] != "bilinear" or up.config["size"] != 2 or \
                self.consumers[up.name] != [head.name]:
            return None
        return up.name if g.layers[up.inbound[0]].out.c % 32 == 0 else None

    def _fused_final_stage(self):
        """The FPN decoder's final_stage, a 3x3 convolution of Concatenate([s2, up2(s3), up4(s4), up8(s5)])
        (nearest up-sampling), in the form that needs none of the up-sampled tensors.  A 3x3 convolution of a
        nearest-x2 up-sampled tensor is, per output phase, a 2x2 convolution of the tensor itself whose taps
        are sums of the 3x3 taps that fall on the same source pixel: the four phase convolutions of the
        transpose-convolution kernels.  So the layer is conv3x3(s2; W[..., :c0]) + T2([s3, up2(s4), up4(s5)]
        at half resolution; phase weights A W[..., c0:]) in the forward pass, and in the backward pass the
        gradient of the phase weights goes back through A^T: 58 % of the multiply-adds in all three
        convolutions of the step.  Returns the layer names involved, or None."""
        g = self.g
        if os.environ.get("MISEG_NO_FUSED_FINAL_STAGE"):
            return None
        for l in g.layers.values():
            if l.cls != "Concatenate" or len(l.inbound) != 4 or len(self.consumers[l.name]) != 1:
                continue
            conv = g.layers[self.consumers[l.name][0]]
            ups = [g.layers[i] for i in l.inbound[1:]]
            fine = g.layers[l.inbound[0]]
            if conv.cls != "Conv2D" or conv.config["kernel_size"] != 3 or conv.config["strides"] != 1 or \
                    conv.config["padding"] != "same" or conv.config["use_bias"] or \
                    conv.config.get("activation") == "relu" or conv.name == g.logits_layer:
                continue
            if any(u.cls != "UpSampling2D" or u.config["interpolation"] != "nearest" or
                   self.consumers[u.name] != [l.name] for u in ups):
                continue
            if [u.config["size"] for u in ups] != [2, 4, 8] or fine.out.c % 64 or \
                    any(u.out.c != fine.out.c for u in ups) or fine.out.h % 2 or fine.out.w % 2 or conv.out.c % 64:
                continue
            return {"cat": l.name, "conv": conv.name, "fine": fine.name, "ups": [u.name for u in ups],
                    "srcs": [u.inbound[0] for u in ups]}
        return None

    def _folded_finest_level(self):
        """The FPN decoder's finest pyramid level, p2 = Add(up2(p3), lateral(skip) + bias), whose only consumer
        is a 3x3 convolution (segm_stage2a): by linearity, zero padding included,
            conv3x3(W3; p2) = T2(q3; A W3) + conv3x3(Wc; skip),  q3 = p3 + bias,  Wc[t] = W_l W3[t],
        the phase convolution of _fused_final_stage on the level above plus a 3x3 convolution of the skip tensor
        with the composed kernel.  p2 (256 channels at half resolution), its 1x1 convolution, the up-sampling
        and the Add never run; the backward applies the product rule to Wc and A^T to the phase weights.
        Returns the layer names involved, or None."""
        g = self.g
        if os.environ.get("MISEG_NO_FOLDED_P2"):
            return None
        for add in g.layers.values():
            if add.cls != "Add" or len(add.inbound) != 2 or len(self.consumers[add.name]) != 1:
                continue
            conv = g.layers[self.consumers[add.name][0]]
            a, b = (g.layers[i] for i in add.inbound)
            up, lat = (a, b) if a.cls == "UpSampling2D" else (b, a)
            if conv.cls != "Conv2D" or conv.config["kernel_size"] != 3 or conv.config["strides"] != 1 or \
                    conv.config["padding"] != "same" or conv.config["use_bias"] or conv.name in self.frozen or \
                    conv.config.get("activation") == "relu" or conv.name == g.logits_layer or conv.out.c % 64:
                continue
            if up.cls != "UpSampling2D" or up.config["interpolation"] != "nearest" or up.config["size"] != 2 or \
                    self.consumers[up.name] != [add.name] or up.out.c % 64:
                continue
            if lat.cls != "Conv2D" or lat.config["kernel_size"] != 1 or lat.config["strides"] != 1 or \
                    not lat.config["use_bias"] or lat.config.get("activation") == "relu" or \
                    self.consumers[lat.name] != [add.name] or lat.name in self.frozen:
                continue
            skip = self._src(lat.inbound[0])[0]
            if g.layers[skip].out.c % 32 or g.layers[skip].cls == "InputLayer":
                continue
            return {"add": add.name, "conv": conv.name, "up": up.name, "lat": lat.name, "skip": skip,
                    "coarse": up.inbound[0]}
        return None

    def _p2_weights(self):
        """Operands of the folded level from the bf16 working copies, once per step: the composed kernel
        [co][3][3][cs], the phase weights [4][co][2][2][cp], their data-gradient packs, and the two factors
        (fp32 values of the bf16 copies) the product rule needs."""
        p2 = self.p2
        cached = self._dgrad_cache.get(p2["conv"])
        if cached is not None:
            return cached
        sg3, sgl = self.segs[(p2["conv"], "kernel")], self.segs[(p2["lat"], "kernel")]
        w3 = self.PB[sg3.off:sg3.off + sg3.count].view(sg3.shape).float()          # [co][3][3][cp]
        cp = self.g.layers[p2["coarse"]].out.c
        cs = self.g.layers[p2["skip"]].out.c
        wl = self.PB[sgl.off:sgl.off + sgl.count].view(sgl.shape).float()[:cp, 0, 0, :cs]   # [cp][cs]
        w3 = w3[..., :cp]
        co = w3.shape[0]
        wc = torch.einsum("otc,cs->ots", w3.reshape(co, 9, cp), wl).to(torch.bfloat16).contiguous()   # [co][9][cs]
        wp32 = torch.einsum("pt,otc->poc", self._t2_matrix(), w3.reshape(co, 9, cp))
        wp32 = wp32.view(4, 4, co, cp).permute(0, 2, 1, 3).contiguous()                      # [4][co][2*2][cp]
        wp = wp32.to(torch.bfloat16)
        wc_d = wc.view(co, 3, 3, cs).float().flip(1, 2).permute(3, 1, 2, 0).to(torch.bfloat16).contiguous()
        wp_d = wp.view(4, co, 2, 2, cp).permute(0, 4, 2, 3, 1).contiguous()
        self._dgrad_cache[p2["conv"]] = (wc, wp, wc_d, wp_d, w3.reshape(co, 9, cp), wl, wp32)
        return self._dgrad_cache[p2["conv"]]

    def _p2_forward(self, l):
        p2, st, n = self.p2, self.stream(), self.n
        p3, skip = self.acts[p2["coarse"]], self.acts[p2["skip"]]
        co, cp = l.out.c, p3.c
        wc, wp = self._p2_weights()[:2]
        # q3 = p3 + the lateral convolution's bias (constant under nearest up-sampling, and inside the zero
        # padding's border like the tensor it is added to)
        q3 = self.new(p3.h, p3.w, cp)
        torch.add(p3.t[..., p3.c0:p3.c0 + cp], self.view(p2["lat"], "bias"), out=q3.t)   # fp32 sum, stored bf16
        part = self.new(skip.h, skip.w, co)
        self.conv(q3, wp.data_ptr(), co, 2, 2, 1, 1, 0, 0, part, q3.h, q3.w, transposed=True)
        out = self.new_out(l.name, skip.h, skip.w, co)
        self.conv(skip, wc.data_ptr(), co, 3, 3, 1, 1, 1, 1, out, skip.h, skip.w, residual=part,
                  stats=self._stats_for(l.name, co))
        self.acts[l.name] = out
        self.saved[l.name] = (skip, q3)

    def _p2_backward(self, l, dy, grads):
        p2, lib, st, n = self.p2, self.lib, self.stream(), self.n
        skip, q3 = self.saved[l.name]
        co, cp, cs = l.out.c, q3.c, skip.c
        wc, wp, wc_d, wp_d, w3, wl, wp32 = self._p2_weights()
        dy_str = (dy.pix, dy.pix * dy.w, dy.pix * dy.w * dy.h)
        csp = _c32(cs)
        gc = torch.zeros((co, 9, csp), dtype=torch.float32, device=self.dev)      # the kernels accumulate
        gp = torch.zeros((4, co, 4, cp), dtype=torch.float32, device=self.dev)
        self.wgrad(skip, dy.ptr, dy_str, gc.data_ptr(), csp, co, 3, 3, 1, 1, 1, 1, skip.h, skip.w, dy_t=dy.t,
                   side=False)
        for a in range(2):
            for b in range(2):
                dyp = dy.ptr + 2 * (a * dy.pix * dy.w + b * dy.pix)
                self.wgrad(q3, dyp, (2 * dy.pix, 2 * dy.pix * dy.w, dy.pix * dy.w * dy.h),
                           gp.data_ptr() + 4 * (a * 2 + b) * co * 4 * cp, cp, co, 2, 2, 1, 1, 1 - a, 1 - b,
                           q3.h, q3.w, dy_t=dy.t, side=False)
        gc = gc[..., :cs]
        # product rule on Wc[o][t][s] = sum_c W3[o][t][c] Wl[c][s]; A^T on the phase weights
        g3 = torch.einsum("ots,cs->otc", gc, wl) + \
            torch.einsum("pt,poc->otc", self._t2_matrix(), gp.permute(0, 2, 1, 3).reshape(16, co, cp))
        self.gview(l.name, "kernel")[..., :cp] = g3.view(co, 3, 3, cp)
        self.gview(p2["lat"], "kernel")[:cp, 0, 0, :cs] = torch.einsum("otc,ots->cs", w3, gc)
        dyv = Act(dy.t, co, dy.c0)
        dq3 = self.new(q3.h, q3.w, cp)
        self.conv(dyv, wp_d.data_ptr(), cp, 2, 2, 1, 1, 0, 0, dq3, q3.h, q3.w, transposed=2)
        # bias gradient = the sum of dq3 over all pixels, taken before dq3 is rounded to bf16 (its entries cancel):
        # sum_pix dq3[c] = sum_{phase, tap, o} Wp[phase][o][tap][c] S[phase][o][tap], S = the sums of dY over the
        # output pixels whose source pixel for that tap lies inside the image -- which is what the phase weight
        # gradient of an all-ones input channel holds (fp32 accumulation, the border handled by its zero padding)
        key = (q3.h, q3.w)
        if self._ones.get("key") != key:
            self._ones = {"key": key, "t": torch.ones((n, q3.h, q3.w, 32), dtype=torch.bfloat16, device=self.dev)}
        ones = Act(self._ones["t"], 32)
        gs = torch.zeros((4, co, 4, 32), dtype=torch.float32, device=self.dev)
        for a in range(2):
            for b in range(2):
                dyp = dy.ptr + 2 * (a * dy.pix * dy.w + b * dy.pix)
                self.wgrad(ones, dyp, (2 * dy.pix, 2 * dy.pix * dy.w, dy.pix * dy.w * dy.h),
                           gs.data_ptr() + 4 * (a * 2 + b) * co * 4 * 32, 32, co, 2, 2, 1, 1, 1 - a, 1 - b,
                           q3.h, q3.w, dy_t=dy.t, side=False)
        self.gview(p2["lat"], "bias").copy_(torch.einsum("pokc,pok->c", wp32, gs[..., 0]))
        if self.grad_log is not None:
            self.grad_log[p2["add"]] = (dy, dq3)       # for checks: the gradient at the level above
        self._accum(grads, p2["coarse"], dq3)

        def run(res):
            out = self.new(skip.h, skip.w, cs)
            self.conv(dyv, wc_d.data_ptr(), cs, 3, 3, 1, 1, 1, 1, out, skip.h, skip.w, residual=res)
            return out
        self._fused_dgrad(grads, p2["skip"], l.name, dy, run)

    # phase (a, b), tap (dy, dx) of the 2x2 form <- the 3x3 taps (ky, kx) it sums
    _T2_ROWS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}

    def _t2_matrix(self):
        """A[p][t]: 1 where 3x3 tap t = ky * 3 + kx falls on source tap p = ((a * 2 + b) * 2 + dy) * 2 + dx."""
        if getattr(self, "_t2_a", None) is None:
            a_ = torch.zeros((16, 9))
            for a in range(2):
                for b in range(2):
                    for dy in range(2):
                        for dx in range(2):
                            for ky in self._T2_ROWS[a][dy]:
                                for kx in self._T2_ROWS[b][dx]:
                                    a_[((a * 2 + b) * 2 + dy) * 2 + dx, ky * 3 + kx] = 1.0
            self._t2_a = a_.to(self.dev)
        return self._t2_a

    def _final_stage_weights(self):
        """bf16 operands of the split final_stage from the master copy, once per step: the fine part
        [co][3][3][c0], the phase weights [4][co][2][2][cc] and the two data-gradient packs."""
        fs = self.fs
        cached = self._dgrad_cache.get(fs["conv"])
        if cached is not None:
            return cached
        c0 = self.g.layers[fs["fine"]].out.c
        k = self.view(fs["conv"], "kernel")                               # [co][3][3][c0 + cc] fp32
        co, cc = k.shape[0], k.shape[3] - c0
        fine = k[..., :c0].to(torch.bfloat16).contiguous()
        wp = torch.einsum("pt,otc->poc", self._t2_matrix(), k[..., c0:].reshape(co, 9, cc))   # [16][co][cc]
        wp = wp.view(4, 4, co, cc).permute(0, 2, 1, 3).to(torch.bfloat16).contiguous()        # [4][co][2*2][cc]
        fine_d = fine.float().flip(1, 2).permute(3, 1, 2, 0).to(torch.bfloat16).contiguous()  # [c0][3][3][co]
        wp_d = wp.view(4, co, 2, 2, cc).permute(0, 4, 2, 3, 1).contiguous()                   # [4][cc][2][2][co]
        self._dgrad_cache[fs["conv"]] = (fine, wp, fine_d, wp_d)
        return self._dgrad_cache[fs["conv"]]

    def _final_stage_forward(self, l):
        fs, lib, st, n = self.fs, self.lib, self.stream(), self.n
        s2 = self.acts[fs["fine"]]
        c0, co = s2.c, l.out.c
        half = self.new(s2.h // 2, s2.w // 2, 3 * c0)
        for i, src in enumerate(fs["srcs"]):
            x = self.acts[src]
            dst = half.slice(i * c0, c0)
            L.check(lib.miseg_upsample_nearest_fwd(x.ptr, None, dst.ptr, n, x.h, x.w, c0, i, x.pix, 0, dst.pix, st))
        fine, wp, _, _ = self._final_stage_weights()
        part = self.new(s2.h, s2.w, co)
        self.conv(half, wp.data_ptr(), co, 2, 2, 1, 1, 0, 0, part, half.h, half.w, transposed=True)
        out = self.new_out(l.name, s2.h, s2.w, co)
        self.conv(s2, fine.data_ptr(), co, 3, 3, 1, 1, 1, 1, out, s2.h, s2.w, residual=part,
                  stats=self._stats_for(l.name, co))
        self.acts[l.name] = out
        self.saved[l.name] = (s2, half)

    def _final_stage_backward(self, l, dy, grads):
        fs, lib, st, n = self.fs, self.lib, self.stream(), self.n
        s2, half = self.saved[l.name]
        c0, co, cc = s2.c, l.out.c, half.c
        fine, wp, fine_d, wp_d = self._final_stage_weights()
        dy_str = (dy.pix, dy.pix * dy.w, dy.pix * dy.w * dy.h)
        # weight gradients of the two parts in their own layouts (main stream: they are folded into the
        # layer's gradient segment right below)
        gf = torch.zeros((co, 3, 3, c0), dtype=torch.float32, device=self.dev)    # the kernels accumulate
        gp = torch.zeros((4, co, 4, cc), dtype=torch.float32, device=self.dev)
        self.wgrad(s2, dy.ptr, dy_str, gf.data_ptr(), c0, co, 3, 3, 1, 1, 1, 1, s2.h, s2.w, dy_t=dy.t, side=False)
        for a in range(2):
            for b in range(2):
                dyp = dy.ptr + 2 * (a * dy.pix * dy.w + b * dy.pix)
                self.wgrad(half, dyp, (2 * dy.pix, 2 * dy.pix * dy.w, dy.pix * dy.w * dy.h),
                           gp.data_ptr() + 4 * (a * 2 + b) * co * 4 * cc, cc, co, 2, 2, 1, 1, 1 - a, 1 - b,
                           half.h, half.w, dy_t=dy.t, side=False)
        gk = self.gview(l.name, "kernel")                                 # [co][3][3][c0 + cc]
        gk[..., :c0] = gf
        gk[..., c0:] = torch.einsum("pt,poc->otc", self._t2_matrix(),
                                    gp.permute(0, 2, 1, 3).reshape(16, co, cc)).view(co, 3, 3, cc)
        # data gradients: the fine part as any 3x3 convolution, the coarse part in one launch of the
        # transpose-convolution data-gradient mode
        dyv = Act(dy.t, co, dy.c0)

        def run(res):
            out = self.new(s2.h, s2.w, c0)
            self.conv(dyv, fine_d.data_ptr(), c0, 3, 3, 1, 1, 1, 1, out, s2.h, s2.w, residual=res)
            return out
        self._fused_dgrad(grads, fs["fine"], l.name, dy, run)
        dhalf = self.new(half.h, half.w, cc)
        self.conv(dyv, wp_d.data_ptr(), cc, 2, 2, 1, 1, 0, 0, dhalf, half.h, half.w, transposed=2)
        for i, src in enumerate(fs["srcs"]):
            x = self.acts[src]
            part = dhalf.slice(i * c0, c0)
            if i == 0:
                self._accum(grads, src, part)
                continue
            dx = self.new(x.h, x.w, c0)
            L.check(lib.miseg_upsample_nearest_bwd(part.ptr, dx.ptr, n, x.h, x.w, c0, i, part.pix, dx.pix, st))
            self._accum(grads, src, dx)

    def _src(self, name):
        """Skip ZeroPadding2D: returns (producer layer name, accumulated padding)."""
        pad = 0
        l = self.g.layers[name]
        while l.cls == "ZeroPadding2D":
            pad += l.config["padding"]
            l = self.g.layers[l.inbound[0]]
        return l.name, pad

    # ------------------------------------------------------------------ parameters
    def _alloc_params(self, weights):
        g = self.g
        segs = OrderedDict()      # (layer, weight) -> Seg
        off = 0

        def add(key, count, shape):
            nonlocal off
            segs[key] = Seg(off, count, shape)
            off += (count + 63) // 64 * 64       # 256-byte aligned segments

        self.conv_info = {}
        for l in g.layers.values():
            if l.cls == "Conv2D":
                kh, kw, ci, co = l.weights[0][1]
                co8 = (co + 7) // 8 * 8          # the weight-gradient kernel writes whole groups of 8 rows
                if kh == 7:
                    shape = (co8, 7, 1, 32)
                else:
                    shape = (co8, kh, kw, _c32(ci))
                add((l.name, "kernel"), int(np.prod(shape)), shape)
                if l.config["use_bias"]:
                    add((l.name, "bias"), co, (co,))
            elif l.cls == "Conv2DTranspose":
                _, _, co, ci = l.weights[0][1]
                shape = (4, co, 2, 2, _c32(ci))
                add((l.name, "kernel"), int(np.prod(shape)), shape)
            elif l.cls == "BatchNormalization":
                c = l.out.c
                if l.config["scale"]:
                    add((l.name, "gamma"), c, (c,))
                add((l.name, "beta"), c, (c,))
        self.segs = segs
        self.nparam = off
        dev = self.dev
        self.P = torch.zeros(off, dtype=torch.float32, device=dev)
        self.G = torch.zeros(off, dtype=torch.float32, device=dev)
        self.M = torch.zeros(off, dtype=torch.float32, device=dev)
        self.V = torch.zeros(off, dtype=torch.float32, device=dev)
        self.PB = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        # non-trainable moving statistics and per-step BN scratch
        self.moving = {}
        self.bn_state = {}
        for l in g.layers.values():
            if l.cls == "BatchNormalization":
                c = l.out.c
                self.moving[l.name] = (torch.zeros(c, device=dev), torch.ones(c, device=dev))
                cp = max(c, 8)
                self.bn_state[l.name] = {k: torch.zeros(cp, device=dev) for k in ("mean", "var", "scale", "shift")}
                self.bn_state[l.name]["ws"] = torch.zeros(2 * cp, dtype=torch.float64, device=dev)
                # statistics gathered by the producing convolution's epilogue: STAT_REPLICAS slots the
                # blocks spread their atomics over, plus one for the totals
                self.bn_state[l.name]["fused"] = torch.zeros((STAT_REPLICAS + 1) * 2 * _c32(c),
                                                             dtype=torch.float64, device=dev)
        self.stem_mask = None
        self.set_weights(weights)
        self.layer_lo = {}
        for (lname, _), sg in segs.items():
            self.layer_lo[lname] = min(self.layer_lo.get(lname, sg.off), sg.off)
        self.buckets = None
        if self.world > 1:
            from .shard import GradBuckets
            self.buckets = GradBuckets(self.G, self.bucket_bytes)

    def view(self, layer, wname):
        s = self.segs[(layer, wname)]
        return self.P[s.off:s.off + s.count].view(s.shape)

    def gview(self, layer, wname):
        s = self.segs[(layer, wname)]
        return self.G[s.off:s.off + s.count].view(s.shape)

    def wptr(self, layer, wname="kernel"):
        return self.PB.data_ptr() + 2 * self.segs[(layer, wname)].off

    def set_weights(self, weights):
        """weights: {layer: {name: np.ndarray}} in Keras layouts."""
        for l in self.g.layers.values():
            if l.name not in weights:
                continue
            w = weights[l.name]
            if l.cls == "Conv2D":
                k = torch.from_numpy(np.ascontiguousarray(w["kernel"], dtype=np.float32)).to(self.dev)
                kh, kw, ci, co = k.shape
                v = self.view(l.name, "kernel")
                v.zero_()
                if kh == 7:
                    v5 = v.view(-1, 7, 1, 8, 4)
                    v5[:co, :, 0, :7, :3] = k.permute(3, 0, 1, 2)
                    if self.stem_mask is None:
                        m = torch.zeros_like(v5)
                        m[:co, :, 0, :7, :3] = 1
                        self.stem_mask = (l.name, m.view(v.shape))
                else:
                    v[:co, :, :, :ci] = k.permute(3, 0, 1, 2)
                if "bias" in w:
                    self.view(l.name, "bias").copy_(torch.from_numpy(np.asarray(w["bias"], np.float32)))
            elif l.cls == "Conv2DTranspose":
                k = torch.from_numpy(np.ascontiguousarray(w["kernel"], dtype=np.float32)).to(self.dev)
                ci = k.shape[3]
                v = self.view(l.name, "kernel")
                v.zero_()
                for a in range(2):
                    for b in range(2):
                        for dy in range(2):
                            for dx in range(2):
                                v[a * 2 + b, :, dy, dx, :ci] = k[3 - a - 2 * dy, 3 - b - 2 * dx]
            elif l.cls == "BatchNormalization":
                if "gamma" in w:
                    self.view(l.name, "gamma").copy_(torch.from_numpy(np.asarray(w["gamma"], np.float32)))
                self.view(l.name, "beta").copy_(torch.from_numpy(np.asarray(w["beta"], np.float32)))
                mm, mv = self.moving[l.name]
                mm.copy_(torch.from_numpy(np.asarray(w["moving_mean"], np.float32)))
                mv.copy_(torch.from_numpy(np.asarray(w["moving_variance"], np.float32)))
        self.PB.copy_(self.P)
        self._dgrad_cache = {}

    def get_weights(self):
        """Back to {layer: {name: np.ndarray}} in Keras layouts."""
        out = OrderedDict()
        for l in self.g.layers.values():
            if not l.weights:
                continue
            d = OrderedDict()
            if l.cls == "Conv2D":
                kh, kw, ci, co = l.weights[0][1]
                v = self.view(l.name, "kernel")
                if kh == 7:
                    k = v.view(-1, 7, 1, 8, 4)[:co, :, 0, :7, :3].permute(1, 2, 3, 0)
                else:
                    k = v[:co, :, :, :ci].permute(1, 2, 3, 0)
                d["kernel"] = k.contiguous().cpu().numpy()
                if l.config["use_bias"]:
                    d["bias"] = self.view(l.name, "bias").cpu().numpy().copy()
            elif l.cls == "Conv2DTranspose":
                _, _, co, ci = l.weights[0][1]
                v = self.view(l.name, "kernel")
                k = torch.zeros((4, 4, co, ci), device=self.dev)
                for a in range(2):
                    for b in range(2):
                        for dy in range(2):
                            for dx in range(2):
                                k[3 - a - 2 * dy, 3 - b - 2 * dx] = v[a * 2 + b, :, dy, dx, :ci]
                d["kernel"] = k.cpu().numpy()
            elif l.cls == "BatchNormalization":
                if l.config["scale"]:
                    d["gamma"] = self.view(l.name, "gamma").cpu().numpy().copy()
                d["beta"] = self.view(l.name, "beta").cpu().numpy().copy()
                d["moving_mean"] = self.moving[l.name][0].cpu().numpy().copy()
                d["moving_variance"] = self.moving[l.name][1].cpu().numpy().copy()
            out[l.name] = d
        return out

    # ------------------------------------------------------------------ kernel wrappers
    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def new(self, h, w, c, zero=False):
        cp = _c32(c)
        if zero or cp != c:
            t = torch.zeros((self.n, h, w, cp), dtype=torch.bfloat16, device=self.dev)
        else:
            t = torch.empty((self.n, h, w, cp), dtype=torch.bfloat16, device=self.dev)
        return Act(t, c)

    def new_out(self, key, h, w, c):
        """Output tensor of layer `key`: a slice of the concatenated buffer it feeds, if any."""
        c0, root = 0, key
        while root in self.place:
            root, off = self.place[root]
            c0 += off
        if self.g.layers[root].cls != "Concatenate":
            return self.new(h, w, c)
        buf = self.concat_bufs.get(root)
        if buf is None:
            rl = self.g.layers[root].out
            buf = self.concat_bufs[root] = self.new(rl.h, rl.w, rl.c)
        return buf.slice(c0, c)

    def conv(self, x, wptr, cout, kh, kw, sy, sx, py, px, out, ho, wo, bias=None, transposed=False,
             out_view=None, act=L.ACT_NONE, probs=None, temperature=1.0, residual=None, stats=None,
             in_valid_c=0, bn_mask=None):
        """out_view: (ptr, pix, row, img) to address a strided lattice; default = dense `out`.
        stats: zeroed fp64 buffer of (STAT_REPLICAS + 1) * 2 * _c32(cout) entries that receives the
        per-channel sum and sum of squares of the stored output (for a following BatchNormalization).
        bn_mask: (x, scale, shift, channels, stats) -- this call is the data gradient of the layer that reads
        relu(x * scale + shift): its first `channels` outputs are zeroed where the ReLU was inactive and their
        sums for the BatchNormalization's backward pass go to `stats` (miseg_conv_desc.res_mode)."""
        d = L.ConvDesc()
        d.inp, d.weight = x.ptr, wptr
        d.N, d.H, d.W, d.Cin = x.n, x.h, x.w, _c32(x.c)
        assert x.c0 + (in_valid_c or d.Cin) <= x.pix, "input view lacks channel padding"
        d.in_valid_c = in_valid_c
        d.Ho, d.Wo, d.Cout = ho, wo, cout
        d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = kh, kw, sy, sx, py, px
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = x.pix, x.pix * x.w, x.pix * x.w * x.h
        if x.pro is not None:
            d.in_scale, d.in_shift = x.pro[0].data_ptr(), x.pro[1].data_ptr()
        if transposed == 2:                       # data gradient of a transpose conv: x is dY (2ho x 2wo)
            assert (x.h, x.w) == (2 * ho, 2 * wo)
            d.H, d.W = ho, wo
        if probs is not None:
            d.out = probs.data_ptr()
            c = probs.shape[-1]
            d.out_pix_stride, d.out_row_stride, d.out_img_stride = c, c * wo, c * wo * ho
            d.out_dtype = L.DT_F32
        elif out_view is not None:
            d.out, d.out_pix_stride, d.out_row_stride, d.out_img_stride = out_view
        else:
            oh, ow = (2 * ho, 2 * wo) if transposed == 1 else (ho, wo)
            assert (out.h, out.w) == (oh, ow), (out.h, out.w, oh, ow)
            d.out = out.ptr
            d.out_pix_stride, d.out_row_stride, d.out_img_stride = out.pix, out.pix * ow, out.pix * ow * oh
        d.shift = bias
        if residual is not None:
            r = residual
            d.residual, d.res_dtype = r.ptr, L.DT_BF16
            d.res_pix_stride, d.res_row_stride, d.res_img_stride = r.pix, r.pix * r.w, r.pix * r.w * r.h
        if bn_mask is not None:
            assert residual is None and bias is None and stats is None
            bx, bscale, bshift, bc, bstats, _ = bn_mask
            d.residual, d.res_dtype = bx.ptr, L.DT_BF16
            d.res_pix_stride, d.res_row_stride, d.res_img_stride = bx.pix, bx.pix * bx.w, bx.pix * bx.w * bx.h
            d.scale, d.shift = bscale, bshift
            d.res_mode, d.res_mask_c = L.RES_BN_RELU_MASK, bc
            stats = (bstats.data_ptr(), _c32(bc))
        d.act = act
        d.transposed = int(transposed)          # True/1: transpose conv; 2: its data gradient
        d.temperature = temperature
        if isinstance(stats, tuple):              # (address, stride): a channel slice of a wider statistics buffer
            d.stats, d.stats_replicas, d.stats_stride = stats[0], STAT_REPLICAS, stats[1]
        elif stats is not None:
            d.stats, d.stats_replicas, d.stats_stride = stats.data_ptr(), STAT_REPLICAS, _c32(cout)
        if bn_mask is not None:
            # the mask is built into the kernels the large layers run on; a layer on another kernel declines
            self._note_conv(d, x, False, True)
            rc = self.lib.miseg_conv2d_fwd(C.byref(d), self.stream())
            if rc == L.E_UNSUPPORTED_SHAPE:
                return False
            L.check(rc)
            return True
        self._note_conv(d, x, probs is not None, residual is not None)
        L.check(self.lib.miseg_conv2d_fwd(C.byref(d), self.stream()))

    def _note_conv(self, d, x, f32_out, has_res):
        """Algorithmic work of a convolution call, for profile_step()."""
        if not self.lib.on:
            return
        buf = C.create_string_buffer(128)
        L.check(self.lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
        taps = d.KH * d.KW * (4 if d.transposed else 1)
        opix = d.N * d.Ho * d.Wo * (4 if d.transposed == 1 else 1)
        macs = d.N * d.Ho * d.Wo * d.Cout * taps * d.Cin
        nbytes = x.n * x.h * x.w * (d.in_valid_c or d.Cin) * 2 + d.Cout * taps * d.Cin * 2 + \
            opix * d.Cout * ((4 if f32_out else 2) + (2 if has_res else 0))
        self.lib.note(buf.value.decode(), macs, nbytes)

    def _stats_for(self, name, cout):
        """The statistics buffer of the one BatchNormalization among the consumers of tensor `name`
        (zeroed, and remembered so that the BatchNormalization only finalises), else None."""
        bns = [c for c in self.consumers.get(name, []) if self.g.layers[c].cls == "BatchNormalization"]
        if len(bns) != 1 or cout % 8 or os.environ.get("MISEG_NO_FUSED_STATS"):
            return None
        bn = self.g.layers[bns[0]]
        if bn.name in self.frozen or bn.name not in self.bn_state:
            return None
        buf = self.bn_state[bn.name]["fused"]
        buf.zero_()
        self.fused_stats.add(bn.name)
        return buf

    def _concat_root(self, name):
        """(outermost Concatenate buffer that tensor `name` lives in, channel offset there) or (None, 0)."""
        c0, root = 0, name
        while root in self.place:
            root, off = self.place[root]
            c0 += off
        if root == name or self.g.layers[root].cls != "Concatenate":
            return None, 0
        return root, c0

    def _cat_stats(self, root):
        """Per-channel sum / sum of squares of everything written into concatenated buffer `root` so far
        (DenseNet: every layer normalises the whole prefix of its block again; the batch statistics of a
        channel do not change once it is written, so they are gathered once, by the epilogue of the
        convolution that produces the channel).  Layout as the per-layer buffers: STAT_REPLICAS + 1 slots
        of [2][stride] doubles."""
        buf = self.cat_stats.get(root)
        if buf is None:
            stride = _c32(self.g.layers[root].out.c)
            buf = self.cat_stats[root] = torch.zeros((STAT_REPLICAS + 1, 2, stride), dtype=torch.float64,
                                                     device=self.dev)
        return buf

    def _cat_stats_for(self, name, cout):
        """Statistics destination for convolution `name` when its output is a channel slice of a concatenated
        buffer whose statistics are being kept (see _cat_stats), else None."""
        root, c0 = self._concat_root(name)
        if root is None or root not in self.cat_seeded or cout % 8 or os.environ.get("MISEG_NO_FUSED_STATS"):
            return None
        buf = self._cat_stats(root)
        return (buf.data_ptr() + 8 * c0, buf.shape[2])

    def _nested_concats(self, root):
        """Names of the Concatenate layers that live at offset 0 of buffer `root` (DenseNet's growing prefix)."""
        return [n for n, (r, off) in self.place.items() if off == 0 and self.g.layers[n].cls == "Concatenate" and
                self._concat_root(n)[0] == root]

    def _grows_by_convs(self, root, first):
        """Is every tensor of buffer `root` other than `first` the output of a convolution whose epilogue can
        gather statistics (no ReLU of its own, a multiple of 8 channels)?"""
        for n, _ in self.place.items():
            if n == first or self._concat_root(n)[0] != root:
                continue
            l = self.g.layers[n]
            if l.cls == "Concatenate":
                continue
            if l.cls != "Conv2D" or l.config.get("activation") == "relu" or l.out.c % 8 or \
                    l.config["kernel_size"] == 7:
                return False
        return True

    def _add_to_fuse(self, name):
        """The Add layer that is the only consumer of convolution `name` and whose other input has been
        computed already (the shortcut of a residual unit), else None."""
        cons = self.consumers.get(name, [])
        if len(cons) != 1 or os.environ.get("MISEG_NO_FUSED_ADD"):
            return None
        add = self.g.layers[cons[0]]
        if add.cls != "Add" or len(add.inbound) != 2 or add.inbound[0] == add.inbound[1]:
            return None
        other = [i for i in add.inbound if i != name][0]
        o = self.acts.get(other)
        if o is None or (o.h, o.w, o.c) != (add.out.h, add.out.w, add.out.c):
            return None
        return add.name

    def wgrad(self, x, dy_ptr, dy_strides, dw_ptr, cin, cout, kh, kw, sy, sx, py, px, ho, wo,
              x_geom=None, dy_t=None, side=True):
        """dy_t: the tensor behind dy_ptr; it is kept from being recycled until the side stream has
        read it."""
        d = L.WgradDesc()
        d.x, d.dy, d.dw = x.ptr, dy_ptr, dw_ptr
        if x_geom is None:
            d.N, d.H, d.W = x.n, x.h, x.w
            d.x_pix, d.x_row, d.x_img = x.pix, x.pix * x.w, x.pix * x.w * x.h
        else:
            d.N, d.H, d.W, d.x_pix, d.x_row, d.x_img = x_geom
        d.Cin, d.Ho, d.Wo, d.Cout = cin, ho, wo, cout
        d.KH, d.KW, d.stride_y, d.stride_x, d.pad_y, d.pad_x = kh, kw, sy, sx, py, px
        d.dy_pix, d.dy_row, d.dy_img = dy_strides
        if getattr(x, "pro", None) is not None:
            d.x_scale, d.x_shift = x.pro[0].data_ptr(), x.pro[1].data_ptr()
        halo = sx == 1 and sy == 1 and kh == kw and kh <= 3 and cin % 32 == 0 and 0 <= py < kh and 0 <= px < kw
        thin = not halo and cin <= 32 and cout <= 64 and 2 <= kh * kw <= 8
        self.lib.note("wgrad_halo_kernel" if halo else ("wgrad_alltaps_kernel" if thin else "conv_wgrad_kernel"),
                      d.N * ho * wo * cout * kh * kw * cin,
                      d.N * d.H * d.W * cin * 2 + d.N * ho * wo * cout * 2 + cout * kh * kw * cin * 4)
        if self.side is None or not side:
            rc = self.lib.miseg_conv2d_wgrad(C.byref(d), self.stream())
        else:
            main = torch.cuda.current_stream(self.dev)
            ev = torch.cuda.Event()
            ev.record(main)                       # dy (and everything before it) is ready
            self.side.wait_event(ev)
            rc = self.lib.miseg_conv2d_wgrad(C.byref(d), self.side.cuda_stream)
        if rc == L.E_UNSUPPORTED_SHAPE and d.x_scale:
            # no kernel with the prologue for this shape: the activated tensor is stored after all
            return self.wgrad(self._stored(x), dy_ptr, dy_strides, dw_ptr, cin, cout, kh, kw, sy, sx, py, px, ho, wo,
                              x_geom=x_geom, dy_t=dy_t, side=side)
        L.check(rc)
        if self.side is None or not side:
            return
        for t in (dy_t, getattr(x, "t", None)):
            if t is not None:
                t.record_stream(self.side)

    def _stored(self, x):
        """The activation of `x` as a tensor in memory: `x` itself, or the BatchNorm + ReLU it still carries applied."""
        if x.pro is None:
            return x
        out = self.new(x.h, x.w, x.c)
        L.check(self.lib.miseg_scale_shift_relu_fwd(x.ptr, out.ptr, x.pro[0].data_ptr(), x.pro[1].data_ptr(), x.pixels,
                                                    x.c, x.pix, out.pix, self.stream()))
        return out

    def _prologue_reader(self, relu_name, xin):
        """The layer that can apply the BatchNorm + ReLU in front of it by itself, if the activated tensor
        `relu_name` is read by exactly one layer and that layer runs on the persistent 32-channel kernel (a 3x3
        stride-1 convolution from 32 stored channels to at most 32, no fused addition): the two full-resolution
        layers at the end of the U-Net decoder, whose apply passes are the largest of the step."""
        if os.environ.get("MISEG_NO_FWD_PROLOGUE") or xin.c != 32 or xin.c0 != 0 or xin.pix != 32:
            return None
        cons = self.consumers.get(relu_name, [])
        if len(cons) != 1:
            return None
        r = self.g.layers[cons[0]]
        if r.cls != "Conv2D" or r.config["kernel_size"] != 3 or r.config["strides"] != 1 or \
                r.config["padding"] != "same" or r.config.get("activation") == "relu":
            return None
        head = r.name == self.g.logits_layer
        if (head and (self.head_up is not None or r.out.c > 4)) or (not head and (r.out.c > 32 or r.out.c % 8)):
            return None
        if not head and self._add_to_fuse(r.name) is not None:
            return None
        if self.fs is not None and r.name == self.fs["conv"] or self.p2 is not None and r.name == self.p2["conv"]:
            return None
        return r.name

    def _join_side(self):
        """The main stream waits for the weight gradients launched so far."""
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)

    def add_into(self, a, b, c=None, out=None):
        """returns a + b (bf16) as a new tensor (or in `out`)."""
        c = c or a.c
        if out is None:
            out = self.new(a.h, a.w, c)
        L.check(self.lib.miseg_copy_channels(a.ptr, b.ptr, out.ptr, a.pixels, c, a.pix, b.pix, out.pix,
                                             self.stream()))
        return out

    def copy(self, src, dst, c):
        L.check(self.lib.miseg_copy_channels(src.ptr, None, dst.ptr, src.pixels, c, src.pix, 0, dst.pix,
                                             self.stream()))

    # ------------------------------------------------------------------ dgrad weight packs
    def dgrad_weight(self, l):
        """bf16 weights of the convolution that maps dY to dX, re-packed from the master copy.
        Returns a list of (weight tensor, kh, kw, taps description) per output phase."""
        key = l.name
        if key in self._dgrad_cache:
            return self._dgrad_cache[key]
        if l.cls == "Conv2D":
            kh, kw, ci, co = l.weights[0][1]
            s = l.config["strides"]
            v = self.view(l.name, "kernel")[:co, :, :, :ci]            # [co][kh][kw][ci]
            cop = _c32(co)

            def pack(sub):                                            # sub: [co][th][tw][ci]
                t = torch.zeros((ci, sub.shape[1], sub.shape[2], cop), dtype=torch.bfloat16, device=self.dev)
                t[..., :co] = sub.permute(3, 1, 2, 0)
                return t
            if s == 1:
                packs = [pack(v.flip(1, 2))]
            else:
                # dX[2y+a, 2x+b] = sum_t dY[y+ty, x+tx] W[ky(a,ty), kx(b,tx)]; pad = (k-1)//2
                p = (kh - 1) // 2
                packs = []
                for a in range(2):
                    for b in range(2):
                        kys = [k for k in range(kh) if (a + p - k) % 2 == 0]
                        kxs = [k for k in range(kw) if (b + p - k) % 2 == 0]
                        # tap offset t = (a + p - k) / 2 ... input row y + t? out row oy = (2y+a+p-k)/2
                        kys = sorted(kys, key=lambda k: (a + p - k) // 2)
                        kxs = sorted(kxs, key=lambda k: (b + p - k) // 2)
                        oys = [(a + p - k) // 2 for k in kys]
                        oxs = [(b + p - k) // 2 for k in kxs]
                        if not kys or not kxs:
                            packs.append(None)
                            continue
                        sub = v[:, kys][:, :, kxs]
                        packs.append((pack(sub), oys[0], oxs[0], len(kys), len(kxs)))
            self._dgrad_cache[key] = packs
            return packs
        # Conv2DTranspose: dIn = 4x4 stride-2 pad-1 convolution of dOut with K[kh][kw][co][ci]
        _, _, co, ci = l.weights[0][1]
        v = self.view(l.name, "kernel")                                # [4][co][2][2][cip]
        k = torch.zeros((ci, 4, 4, _c32(co)), dtype=torch.bfloat16, device=self.dev)
        for a in range(2):
            for b in range(2):
                for dy in range(2):
                    for dx in range(2):
                        k[:, 3 - a - 2 * dy, 3 - b - 2 * dx, :co] = v[a * 2 + b, :, dy, dx, :ci].t()
        packs = [k]
        if _c32(co) % 64 == 0 and co == _c32(co):
            # phase form for the one-launch data gradient (transposed = 2): [phase][ci][dy][dx][co]
            packs.append(v[..., :ci].permute(0, 4, 2, 3, 1).to(torch.bfloat16).contiguous())
        self._dgrad_cache[key] = packs
        return packs

    # ------------------------------------------------------------------ forward (training mode)
    def forward(self, x, momentum_update=True):
        """x: fp32 (n,H,W,3) device tensor.  Returns fp32 probabilities (n,H,W,classes)."""
        g, lib, st = self.g, self.lib, self.stream()
        self.acts = {}
        self.saved = {}
        self.concat_bufs = {}
        self.fused_stats = set()
        self.fused_adds = set()
        self.cat_stats = {}
        self.cat_seeded = set()
        self.fwd_prologues = set()      # BatchNormalizations whose apply pass the reading convolution does
        n = self.n
        self.x_in = x.contiguous()
        for name in self.order:
            l = g.layers[name]
            cls = l.cls
            lib.tag = "fwd " + name
            if cls in ("InputLayer", "ZeroPadding2D"):
                continue
            if cls == "BatchNormalization" and g.layers[l.inbound[0]].cls == "InputLayer":
                # bn_data: batch statistics of the fp32 input; applied by the stem's input preparation
                s = self.bn_state[name]
                if name in self.frozen:
                    self._frozen_bn(l, s)
                    continue
                mm, mv = self.moving[name] if momentum_update else (None, None)
                L.check(lib.miseg_input_stats_fwd(
                    self.x_in.data_ptr(), n * self.H * self.W, self.view(name, "beta").data_ptr(),
                    mm.data_ptr() if mm is not None else None, mv.data_ptr() if mv is not None else None,
                    l.config["momentum"], l.config["epsilon"], s["mean"].data_ptr(), s["var"].data_ptr(),
                    s["scale"].data_ptr(), s["shift"].data_ptr(), s["ws"].data_ptr(), st))
                continue
            if self.fs is not None and (name == self.fs["cat"] or name in self.fs["ups"]):
                continue                           # neither the up-sampled tensors nor their concatenation exist
            if self.fs is not None and name == self.fs["conv"]:
                self._final_stage_forward(l)
                continue
            if self.p2 is not None and name in (self.p2["lat"], self.p2["up"], self.p2["add"]):
                continue                           # folded into the 3x3 convolution that consumes the level
            if self.p2 is not None and name == self.p2["conv"]:
                self._p2_forward(l)
                continue
            if cls == "Conv2D":
                src, pad = self._src(l.inbound[0])
                k = l.config["kernel_size"]
                stride = l.config["strides"]
                if l.config["padding"] == "same":
                    pad = (k - 1) // 2
                co = l.out.c
                bias = self.view(name, "bias").data_ptr() if l.config["use_bias"] else None
                if k == 7:
                    self._stem_forward(l, src)
                    continue
                if g.layers[src].cls == "InputLayer" and src not in self.acts:
                    # a 3x3 convolution on the image itself (VGG block1_conv1): the fp32 image as a
                    # bf16 tensor zero-padded to 32 channels
                    t = torch.empty((n, self.H, self.W, 32), dtype=torch.bfloat16, device=self.dev)
                    L.check(lib.miseg_im2col3x3_fwd(self.x_in.data_ptr(), t.data_ptr(), n, self.H, self.W, 1,
                                                    L.FMT_BF16, st))
                    self.acts[src] = Act(t, 3)
                relu = l.config.get("activation") == "relu"
                if name == g.logits_layer and self.head_up is not None:
                    # rows c * 9 + tap of the head's [class][3][3][cin] weights as a 1x1 convolution at half
                    # resolution (fp32 result), then sampling, tap sum, bias and activation in one kernel
                    xin = self.acts[g.layers[self.head_up].inbound[0]]
                    y = torch.empty((n, xin.h, xin.w, 32), dtype=torch.float32, device=self.dev)
                    self.conv(xin, self.wptr(name), 32, 1, 1, 1, 1, 0, 0, None, xin.h, xin.w, probs=y)
                    probs = torch.empty((n, l.out.h, l.out.w, co), dtype=torch.float32, device=self.dev)
                    act = L.ACT_SIGMOID if g.activation == "sigmoid" else L.ACT_SOFTMAX
                    self.lib.note("head_gather_kernel", 0, y.numel() * 4 + probs.numel() * 4)
                    L.check(lib.miseg_head_gather_fwd(y.data_ptr(), bias, probs.data_ptr(), None, n, xin.h, xin.w,
                                                      co, 32, act, g.temperature if g.temperature else 1.0, 0, 1,
                                                      st))
                    self.probs = probs
                    self.saved[name] = (xin, pad)
                    continue
                xin = self.acts[src]
                if name == g.logits_layer:
                    probs = torch.empty((n, l.out.h, l.out.w, co), dtype=torch.float32, device=self.dev)
                    act = L.ACT_SIGMOID if g.activation == "sigmoid" else L.ACT_SOFTMAX
                    self.conv(xin, self.wptr(name), co, k, k, stride, stride, pad, pad, None, l.out.h,
                              l.out.w, bias=bias, act=act, probs=probs,
                              temperature=g.temperature if g.temperature else 1.0)
                    self.probs = probs
                    self.saved[name] = (xin, pad)
                    continue
                add = self._add_to_fuse(name) if not relu else None
                if add is not None:
                    # Add([this convolution, a tensor that exists already]): the addition happens in the
                    # epilogue (as do the statistics of a BatchNormalization that follows the Add), and the
                    # convolution's own output is never stored
                    other = [i for i in g.layers[add].inbound if i != name][0]
                    out = self.new_out(add, l.out.h, l.out.w, co)
                    self.conv(xin, self.wptr(name), co, k, k, stride, stride, pad, pad, out, l.out.h, l.out.w,
                              bias=bias, residual=self.acts[other], stats=self._stats_for(add, co))
                    self.acts[add] = out
                    self.fused_adds.add(add)
                    self.saved[name] = (xin, pad)
                    continue
                out = self.new_out(name, l.out.h, l.out.w, co)
                stats = None
                if not relu:
                    stats = self._stats_for(name, co)
                    if stats is None:
                        stats = self._cat_stats_for(name, co)
                self.conv(xin, self.wptr(name), co, k, k, stride, stride, pad, pad, out, l.out.h, l.out.w,
                          bias=bias, stats=stats, act=L.ACT_RELU if relu else L.ACT_NONE)
                self.acts[name] = out
                self.saved[name] = (xin, pad)
            elif cls == "Conv2DTranspose":
                xin = self.acts[l.inbound[0]]
                out = self.new_out(name, l.out.h, l.out.w, l.out.c)
                self.conv(xin, self.wptr(name), l.out.c, 2, 2, 1, 1, 0, 0, out, xin.h, xin.w, transposed=True,
                          stats=self._stats_for(name, l.out.c))
                self.acts[name] = out
                self.saved[name] = (xin, 0)
            elif cls == "BatchNormalization":
                xin = self.acts[l.inbound[0]]
                s = self.bn_state[name]
                c = l.out.c
                mm, mv = self.moving[name] if momentum_update else (None, None)
                gamma = self.view(name, "gamma").data_ptr() if l.config["scale"] else None
                if name in self.frozen:
                    self._frozen_bn(l, s)
                elif name in self.fused_stats:
                    L.check(lib.miseg_bn_finalize(
                        s["fused"].data_ptr(), STAT_REPLICAS, _c32(c), xin.pixels, c, gamma,
                        self.view(name, "beta").data_ptr(),
                        mm.data_ptr() if mm is not None else None, mv.data_ptr() if mv is not None else None,
                        l.config["momentum"], l.config["epsilon"], s["mean"].data_ptr(), s["var"].data_ptr(),
                        s["scale"].data_ptr(), s["shift"].data_ptr(), st))
                elif l.inbound[0] in self.cat_seeded:
                    # the input is a concatenated buffer whose channels all brought their statistics along
                    buf = self.cat_stats[l.inbound[0] if l.inbound[0] in self.cat_stats else
                                         self._concat_root(l.inbound[0])[0]]
                    L.check(lib.miseg_bn_finalize(
                        buf.data_ptr(), STAT_REPLICAS, buf.shape[2], xin.pixels, c, gamma,
                        self.view(name, "beta").data_ptr(),
                        mm.data_ptr() if mm is not None else None, mv.data_ptr() if mv is not None else None,
                        l.config["momentum"], l.config["epsilon"], s["mean"].data_ptr(), s["var"].data_ptr(),
                        s["scale"].data_ptr(), s["shift"].data_ptr(), st))
                else:
                    L.check(lib.miseg_bn_stats_fwd(
                        xin.ptr, xin.pixels, c, xin.pix, gamma, self.view(name, "beta").data_ptr(),
                        mm.data_ptr() if mm is not None else None, mv.data_ptr() if mv is not None else None,
                        l.config["momentum"], l.config["epsilon"], s["mean"].data_ptr(), s["var"].data_ptr(),
                        s["scale"].data_ptr(), s["shift"].data_ptr(), s["ws"].data_ptr(), st))
                    root, c0 = self._concat_root(l.inbound[0])
                    if root is not None and c0 == 0 and not os.environ.get("MISEG_NO_FUSED_STATS") and \
                            self._grows_by_convs(root, l.inbound[0]):
                        # the first tensor of a concatenated buffer that convolutions then extend (the input
                        # of a DenseNet block): its statistics, just computed, seed the buffer's
                        buf = self._cat_stats(root)
                        buf.zero_()
                        npx = float(xin.pixels)
                        mean64, var64 = s["mean"][:c].double(), s["var"][:c].double()
                        buf[0, 0, :c] = mean64 * npx
                        buf[0, 1, :c] = (var64 + mean64 * mean64) * npx
                        self.cat_seeded.add(root)
                        for nm in self._nested_concats(root):
                            self.cat_seeded.add(nm)
                cons = self.consumers[name]
                relu = len(cons) == 1 and g.layers[cons[0]].cls == "Activation" and \
                    g.layers[cons[0]].config["activation"] == "relu"
                if not relu:
                    raise NotImplementedError("BatchNormalization without a following ReLU: %s" % name)
                if self._prologue_reader(cons[0], xin) is not None:
                    # the one reader applies scale/shift + ReLU to its operand tiles: no apply pass, no second tensor
                    self.acts[cons[0]] = Act(xin.t, c, xin.c0, pro=(s["scale"], s["shift"]))
                    self.fwd_prologues.add(name)
                else:
                    out = self.new_out(cons[0], xin.h, xin.w, c)
                    L.check(lib.miseg_scale_shift_relu_fwd(xin.ptr, out.ptr, s["scale"].data_ptr(),
                                                           s["shift"].data_ptr(), xin.pixels, c, xin.pix,
                                                           out.pix, st))
                    self.acts[cons[0]] = out          # the ReLU's output
                self.saved[name] = xin
            elif cls == "Activation":
                continue                           # relu: fused with its BatchNormalization; softmax: head
            elif cls == "TemperatureScaledSoftmax":
                continue
            elif cls == "MaxPooling2D":
                src, _ = self._src(l.inbound[0])
                xin = self.acts[src]
                out = self.new_out(name, l.out.h, l.out.w, l.out.c)
                if l.config["pool_size"] == 2:
                    L.check(lib.miseg_maxpool2x2_fwd(xin.ptr, out.ptr, n, xin.h, xin.w, xin.c, xin.pix,
                                                     out.pix, L.FMT_BF16, st))
                else:
                    L.check(lib.miseg_maxpool3x3s2_fwd(xin.ptr, out.ptr, None, None, n, xin.h, xin.w, xin.c,
                                                       xin.pix, out.pix, st))
                self.acts[name] = out
                self.saved[name] = xin
            elif cls == "AveragePooling2D":
                xin = self.acts[l.inbound[0]]
                out = self.new_out(name, l.out.h, l.out.w, l.out.c)
                L.check(lib.miseg_avgpool2x2_fwd(xin.ptr, out.ptr, n, xin.h, xin.w, xin.c, xin.pix, out.pix, st))
                self.acts[name] = out
            elif cls == "UpSampling2D":
                if name == self.head_up:
                    continue                       # sampled inside the head's gather kernel
                xin = self.acts[l.inbound[0]]
                out = self.new_out(name, l.out.h, l.out.w, l.out.c) if self._add_to_fuse(name) is None or \
                    l.config["interpolation"] == "bilinear" else None
                if l.config["interpolation"] == "bilinear":
                    L.check(lib.miseg_upsample_bilinear2x_fwd(xin.ptr, out.ptr, n, xin.h, xin.w, xin.c,
                                                              xin.pix, out.pix, st))
                else:
                    shift = int(round(math.log2(l.config["size"])))
                    add = self._add_to_fuse(name)
                    if add is not None:
                        # Add([this up-sampling, a tensor that exists already]) (FPN's top-down path): the
                        # kernel adds while it copies, and the up-sampled tensor itself is never stored
                        other = self.acts[[i for i in g.layers[add].inbound if i != name][0]]
                        out = self.new_out(add, l.out.h, l.out.w, l.out.c)
                        L.check(lib.miseg_upsample_nearest_fwd(xin.ptr, other.ptr, out.ptr, n, xin.h, xin.w, xin.c,
                                                               shift, xin.pix, other.pix, out.pix, st))
                        self.acts[add] = out
                        self.fused_adds.add(add)
                        continue
                    L.check(lib.miseg_upsample_nearest_fwd(xin.ptr, None, out.ptr, n, xin.h, xin.w, xin.c,
                                                           shift, xin.pix, 0, out.pix, st))
                self.acts[name] = out
            elif cls == "Add":
                if name in self.fused_adds:
                    continue
                a, b = self.acts[l.inbound[0]], self.acts[l.inbound[1]]
                self.acts[name] = self.add_into(a, b, l.out.c, out=self.new_out(name, l.out.h, l.out.w, l.out.c))
            elif cls == "Concatenate":
                out = self.new_out(name, l.out.h, l.out.w, l.out.c)
                c0 = 0
                for i in l.inbound:
                    src = self.acts[i]
                    if self.place.get(i) != (name, c0):          # not produced in place
                        self.copy(src, out.slice(c0, src.c), src.c)
                    c0 += src.c
                self.acts[name] = out
            else:
                raise NotImplementedError("layer class %s" % cls)
        return self.probs

    def _frozen_bn(self, l, s):
        """Inference-mode coefficients from the moving statistics."""
        c = l.out.c
        mm, mv = self.moving[l.name]
        scale = torch.rsqrt(mv + l.config["epsilon"])
        if l.config["scale"]:
            scale = scale * self.view(l.name, "gamma")
        s["scale"][:c] = scale
        s["shift"][:c] = self.view(l.name, "beta") - mm * scale
        s["mean"][:c] = mm
        s["var"][:c] = mv

    def _stem_forward(self, l, src):
        lib, st, n = self.lib, self.stream(), self.n
        h, w = self.H, self.W
        padded = torch.empty((n, h + 6, w + 8, 4), dtype=torch.bfloat16, device=self.dev)
        srcl = self.g.layers[src]
        if srcl.cls == "BatchNormalization":
            s = self.bn_state[src]
            sc, sh = s["scale"], s["shift"]
        else:
            sc = torch.ones(8, device=self.dev)
            sh = torch.zeros(8, device=self.dev)
        # channel 3 = indicator of "inside the image": the stem's weight gradient for that (dead)
        # channel then equals the per-tap sums of dY that the input BatchNorm's beta gradient needs
        L.check(lib.miseg_prep_input_indicator(self.x_in.data_ptr(), padded.data_ptr(), sc.data_ptr(),
                                               sh.data_ptr(), n, h, w, st))
        out = self.new_out(l.name, h // 2, w // 2, 64)
        d = L.ConvDesc()
        d.inp, d.weight = padded.data_ptr(), self.wptr(l.name)
        d.N, d.H, d.W, d.Cin = n, h + 6, w // 2, 32
        d.Ho, d.Wo, d.Cout = h // 2, w // 2, 64
        d.KH, d.KW, d.stride_y, d.stride_x = 7, 1, 2, 1
        d.in_pix_stride, d.in_row_stride, d.in_img_stride = 8, (w + 8) * 4, (h + 6) * (w + 8) * 4
        d.out = out.ptr
        d.out_pix_stride, d.out_row_stride, d.out_img_stride = out.pix, out.pix * (w // 2), out.pix * (w // 2) * (h // 2)
        d.temperature = 1.0
        # no statistics in this convolution's epilogue: without them the launch takes the persistent stem
        # kernel, and that plus a separate statistics pass over its output is faster than the generic
        # kernel with the statistics fused (1.4 against 2.7 ms per 16 images at 1920x2560)
        if self.lib.on:
            buf = C.create_string_buffer(128)
            L.check(lib.miseg_conv_kernel_name(C.byref(d), buf, 128))
            pix = n * (h // 2) * (w // 2)
            self.lib.note(buf.value.decode(), pix * 64 * 147, n * (h + 6) * (w + 8) * 8 + pix * 128)
        L.check(lib.miseg_conv2d_fwd(C.byref(d), st))
        self.acts[l.name] = out
        self.saved[l.name] = (padded, src)

    # ------------------------------------------------------------------ loss
    def loss_and_metrics(self, gt):
        """gt: fp32 one-hot (n,H,W,classes) device tensor.  Returns dict and keeps the sums."""
        c = self.classes
        self.gt = gt.contiguous()
        pixels = self.n * self.H * self.W
        if self.focal:
            self.loss_sums = torch.empty(5 * c, dtype=torch.float64, device=self.dev)
            L.check(self.lib.miseg_loss_sums_focal(self.gt.data_ptr(), self.probs.data_ptr(),
                                                   self.loss_sums.data_ptr(), pixels, c, self.focal_alpha,
                                                   self.focal_gamma, self.stream()))
        else:
            self.loss_sums = torch.empty(4 * c, dtype=torch.float64, device=self.dev)
            L.check(self.lib.miseg_loss_sums(self.gt.data_ptr(), self.probs.data_ptr(),
                                             self.loss_sums.data_ptr(), pixels, c, self.stream()))
        return self.loss_sums, pixels

    # ------------------------------------------------------------------ backward
    def _fused_dgrad(self, grads, src_name, producer, dy, run):
        """Data gradient of a convolution accumulated onto the gradient its input already holds:
        `run(residual)` launches the convolution with that gradient as the epilogue's residual."""
        res = grads.pop(src_name, None)
        if res is None:
            self._accum(grads, src_name, run(None), producer, dy)
            return
        if self.grad_log is not None:            # checks want the un-accumulated part as well
            part = run(None)
            self.grad_log[producer] = (dy, part)
            self.accum_log.setdefault(src_name, []).append(part)
        grads[src_name] = run(res)

    def _accum(self, grads, name, gact, producer=None, dy=None):
        if self.grad_log is not None:
            if producer is not None:
                self.grad_log[producer] = (dy, gact)
            self.accum_log.setdefault(name, []).append(gact)
        if name in grads:
            grads[name] = self.add_into(grads[name], gact, gact.c)
        else:
            grads[name] = gact

    def backward(self, loss_scale=1.0):
        g, lib, st, n = self.g, self.lib, self.stream(), self.n
        self.G.zero_()
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
        grads = {}
        self.bwd_fused = set()        # BatchNormalizations whose backward sums a data gradient's epilogue gathers
        self.bwd_masked = {}          # layer -> leading channels of its data gradient that carry the ReLU's mask
        self.grad_log = {} if self.keep_grads else None
        self.accum_log, self.accum_final = {}, {}
        c = self.classes
        head = g.layers[g.logits_layer]
        # the fused head reads `classes` channels per pixel, and so do a plain 3x3 head's gradient kernels when
        # its input has at most 32 channels and no other consumer (miseg_conv_desc.in_valid_c); otherwise the
        # gradient is a 32-channel tensor like any other
        dzc = 8 if self.head_up is not None or self._narrow_dz(head) else 32
        dz = torch.empty((n, self.H, self.W, dzc), dtype=torch.bfloat16, device=self.dev)   # the kernel zero-pads
        temp = g.temperature if g.temperature else 1.0
        L.check(lib.miseg_loss_grad_ex(self.gt.data_ptr(), self.probs.data_ptr(), self.loss_sums.data_ptr(),
                                       dz.data_ptr(), n * self.H * self.W, c, dzc, loss_scale, temp,
                                       self.loss_flags, self.focal_alpha, self.focal_gamma, st))
        grads[head.name] = Act(dz, c)
        if self.buckets is not None:
            self.buckets.begin()
        for name in reversed(self.order):
            l = g.layers[name]
            cls = l.cls
            lib.tag = "bwd " + name
            if cls in ("InputLayer", "ZeroPadding2D", "TemperatureScaledSoftmax"):
                continue
            if cls == "Activation":
                if l.config["activation"] == "relu" and name in grads:
                    # hand the gradient to the BatchNormalization this ReLU is fused with
                    if self.grad_log is not None:
                        self.accum_final[name] = grads[name]
                    grads[l.inbound[0]] = grads.pop(name)
                continue
            if cls == "BatchNormalization" and g.layers[l.inbound[0]].cls == "InputLayer":
                continue                      # bn_data: handled with the stem
            if name not in grads:
                continue
            if name in self.frozen:
                grads.pop(name)               # nothing upstream of a frozen backbone layer learns
                continue
            dy = grads.pop(name)
            if self.grad_log is not None:
                self.accum_final[name] = dy
            if cls == "Conv2D" and name == g.logits_layer and self.head_up is not None:
                self._fused_head_backward(l, dy, grads)
            elif self.fs is not None and name == self.fs["conv"]:
                self._final_stage_backward(l, dy, grads)
            elif self.p2 is not None and name == self.p2["conv"]:
                self._p2_backward(l, dy, grads)
            elif cls == "Conv2D":
                if l.config.get("activation") == "relu":
                    # Conv2D(activation="relu"): the gradient of the pre-activation
                    y = self.acts[name]
                    pre = self.new(y.h, y.w, y.c)
                    L.check(lib.miseg_relu_bwd(y.ptr, dy.ptr, None, pre.ptr, y.pixels, y.c, y.pix, dy.pix, 0,
                                               pre.pix, st))
                    dy = pre
                self._conv_backward(l, dy, grads)
            elif cls == "Conv2DTranspose":
                self._convT_backward(l, dy, grads)
            elif cls == "BatchNormalization":
                xin = self.saved[name]
                s = self.bn_state[name]
                cc = l.out.c
                sums = torch.empty(2 * cc, dtype=torch.float64, device=self.dev)
                masked = name in self.bwd_fused
                if masked:
                    # dy is already the gradient behind the ReLU and its sums came with it
                    L.check(lib.miseg_bn_bwd_sums(s["fused"].data_ptr(), STAT_REPLICAS, _c32(cc), cc,
                                                  s["mean"].data_ptr(), s["var"].data_ptr(), l.config["epsilon"],
                                                  sums.data_ptr(), st))
                else:
                    L.check(lib.miseg_bn_relu_bwd_reduce(xin.ptr, dy.ptr, xin.pixels, cc, xin.pix, dy.pix,
                                                         s["scale"].data_ptr(), s["shift"].data_ptr(),
                                                         s["mean"].data_ptr(), s["var"].data_ptr(),
                                                         l.config["epsilon"], 1, sums.data_ptr(), st))
                def run(res, xin=xin, s=s, cc=cc, sums=sums, dy=dy, eps=l.config["epsilon"], relu=0 if masked else 1):
                    out = self.new(xin.h, xin.w, cc)
                    L.check(lib.miseg_bn_relu_bwd_apply(
                        xin.ptr, dy.ptr, out.ptr, xin.pixels, cc, xin.pix, dy.pix, out.pix,
                        s["scale"].data_ptr(), s["shift"].data_ptr(), s["mean"].data_ptr(),
                        s["var"].data_ptr(), eps, relu, sums.data_ptr(), res.ptr if res is not None else None,
                        res.pix if res is not None else 0, st))
                    return out
                self.gview(name, "beta").copy_(sums[:cc])
                if l.config["scale"]:
                    self.gview(name, "gamma").copy_(sums[cc:])
                self._fused_dgrad(grads, l.inbound[0], name, dy, run)
            elif cls == "MaxPooling2D":
                xin = self.saved[name]
                if l.config["pool_size"] == 2:
                    dx = self.new(xin.h, xin.w, xin.c)
                    L.check(lib.miseg_maxpool2x2_bwd(xin.ptr, dy.ptr, dx.ptr, n, xin.h, xin.w, xin.c, xin.pix,
                                                     dy.pix, dx.pix, st))
                    self._accum(grads, self._src(l.inbound[0])[0], dx, name, dy)
                else:
                    def run(res, xin=xin, dy=dy):
                        out = self.new(xin.h, xin.w, xin.c)
                        L.check(lib.miseg_maxpool3x3s2_bwd(xin.ptr, dy.ptr, out.ptr, n, xin.h, xin.w, xin.c, xin.pix,
                                                           dy.pix, out.pix, res.ptr if res is not None else None,
                                                           res.pix if res is not None else 0, st))
                        return out
                    # the gradient the pool's input already holds (a skip connection's) is added in the kernel
                    self._fused_dgrad(grads, self._src(l.inbound[0])[0], name, dy, run)
            elif cls == "AveragePooling2D":
                src = g.layers[l.inbound[0]].out
                dx = self.new(src.h, src.w, src.c)
                L.check(lib.miseg_avgpool2x2_bwd(dy.ptr, dx.ptr, n, src.h, src.w, src.c, dy.pix, dx.pix, st))
                self._accum(grads, l.inbound[0], dx, name, dy)
            elif cls == "UpSampling2D":
                src = g.layers[l.inbound[0]].out
                dx = self.new(src.h, src.w, src.c)
                if l.config["interpolation"] == "bilinear":
                    L.check(lib.miseg_upsample_bilinear2x_bwd(dy.ptr, dx.ptr, n, src.h, src.w, src.c, dy.pix,
                                                              dx.pix, st))
                else:
                    shift = int(round(math.log2(l.config["size"])))
                    L.check(lib.miseg_upsample_nearest_bwd(dy.ptr, dx.ptr, n, src.h, src.w, src.c, shift,
                                                           dy.pix, dx.pix, st))
                self._accum(grads, l.inbound[0], dx, name, dy)
            elif cls == "Add":
                for i in l.inbound:
                    self._accum(grads, i, dy)
            elif cls == "Concatenate":
                c0 = 0
                for i in l.inbound:
                    ci = g.layers[i].out.c
                    if ci % 32 == 0 and (dy.c0 + c0) % 8 == 0:
                        part = dy.slice(c0, ci)                   # consumers read through the stride
                    else:
                        part = self.new(dy.h, dy.w, ci)
                        self.copy(dy.slice(c0, ci), part, ci)
                    self._accum(grads, i, part)
                    c0 += ci
            else:
                raise NotImplementedError("backward of %s" % cls)
            self._grads_final(name)
        self._join_side()
        if self.buckets is not None:
            self.buckets.flush()

    def _sole_reader(self, name):
        """The one layer that reads tensor `name`, looking through ZeroPadding2D; None if several do."""
        cons = self.consumers.get(name, [])
        while len(cons) == 1 and self.g.layers[cons[0]].cls == "ZeroPadding2D":
            cons = self.consumers.get(cons[0], [])
        return cons[0] if len(cons) == 1 else None

    def _bn_mask_for(self, src_name, reader, grads):
        """When tensor `src_name` is relu(bn(x)) -- or a Concatenate that starts with it -- and `reader` is the
        only layer that reads it, the reader's data gradient can apply the ReLU's mask and gather the
        BatchNormalization's two backward reductions while it stores (one pass over x and the gradient less
        per BatchNormalization).  Returns the `bn_mask` argument of conv(), or None."""
        if os.environ.get("MISEG_NO_FUSED_BN_BWD") or src_name in grads:
            return None
        g = self.g
        relu = src_name
        if g.layers[relu].cls == "Concatenate":
            if self._sole_reader(relu) != reader:
                return None
            cat = relu
            relu = g.layers[cat].inbound[0]
            if self.consumers.get(relu) != [cat] or self.place.get(relu) != (cat, 0):
                return None
        elif self._sole_reader(relu) != reader:
            return None
        rl = g.layers[relu]
        if rl.cls != "Activation" or rl.config.get("activation") != "relu":
            return None
        bn = g.layers[rl.inbound[0]]
        if bn.cls != "BatchNormalization" or bn.name in self.frozen or bn.name not in self.bn_state or \
                bn.name not in self.saved or bn.out.c % 8 or self.consumers.get(bn.name) != [relu]:
            return None
        s = self.bn_state[bn.name]
        buf = s["fused"]
        buf.zero_()
        self.bwd_fused.add(bn.name)
        return (self.saved[bn.name], s["scale"].data_ptr(), s["shift"].data_ptr(), bn.out.c, buf, bn.name)

    def _masked_dgrad(self, layer, bn_mask, launch):
        """launch(bn_mask) runs a data gradient; with a mask the kernel may decline (conv() returns False): the
        layer then runs plain and its BatchNormalization makes its own reduction pass."""
        if bn_mask is not None:
            if launch(bn_mask) is not False:
                self.bwd_masked[layer] = bn_mask[3]
                return
            self.bwd_fused.discard(bn_mask[5])
        launch(None)

    def _narrow_dz(self, head):
        if head.cls != "Conv2D" or head.config["kernel_size"] != 3 or head.config["strides"] != 1 or \
                os.environ.get("MISEG_WIDE_DZ") or self.classes > 8:
            return False
        src = self._src(head.inbound[0])[0]
        sl = self.g.layers[src]
        return sl.cls != "InputLayer" and _c32(sl.out.c) == 32 and self.consumers[head.inbound[0]] == [head.name] \
            and len(self.consumers[src]) == 1 and head.name not in self.frozen

    def _fused_head_backward(self, l, dz, grads):
        """Backward of final_upsampling + head_conv in their half-resolution form (see _fused_head_source)."""
        lib, st, n = self.lib, self.stream(), self.n
        name, co = l.name, l.out.c
        xin, _ = self.saved[name]
        if l.config["use_bias"]:
            sums = torch.empty(8, dtype=torch.float64, device=self.dev)
            L.check(lib.miseg_channel_sums(dz.ptr, dz.pixels, 8, dz.pix, sums.data_ptr(), st))
            self.gview(name, "bias").copy_(sums[:co])
        dyt = torch.empty((n, xin.h, xin.w, 32), dtype=torch.bfloat16, device=self.dev)
        lib.note("head_gather_bwd_kernel", 0, dz.t.numel() * 2 + dyt.numel() * 2)
        L.check(lib.miseg_head_gather_bwd(dz.ptr, dz.pix, dyt.data_ptr(), n, xin.h, xin.w, co, 32, st))
        dy = Act(dyt, 32)
        # the weight gradient's 32 rows are rows c * 9 + tap of the [class][3][3][cin] segment
        seg = self.segs[(name, "kernel")]
        cin = _c32(xin.c)
        self.wgrad(xin, dy.ptr, (32, 32 * xin.w, 32 * xin.w * xin.h), self.G.data_ptr() + 4 * seg.off, cin, 32,
                   1, 1, 1, 1, 0, 0, xin.h, xin.w, dy_t=dyt)
        wt = self._dgrad_cache.get(name)
        if wt is None:
            rows = self.view(name, "kernel").reshape(-1, cin)[:32]                  # [c * 9 + tap][cin]
            wt = self._dgrad_cache[name] = rows.t().to(torch.bfloat16).contiguous()  # [cin][32]
        src_name = self.g.layers[self.head_up].inbound[0]

        def run(res):
            out = self.new(xin.h, xin.w, xin.c)
            self.conv(dy, wt.data_ptr(), xin.c, 1, 1, 1, 1, 0, 0, out, xin.h, xin.w, residual=res)
            return out
        self._fused_dgrad(grads, src_name, name, dz, run)

    def _conv_backward(self, l, dy, grads):
        lib, st, n = self.lib, self.stream(), self.n
        name = l.name
        k = l.config["kernel_size"]
        stride = l.config["strides"]
        co = l.out.c
        cop = _c32(co)
        if l.config["use_bias"]:
            cs = min(cop, dy.pix)                     # (the gradient at the logits may be 8 channels wide)
            sums = torch.empty(cs, dtype=torch.float64, device=self.dev)
            L.check(lib.miseg_channel_sums(dy.ptr, dy.pixels, cs, dy.pix, sums.data_ptr(), st))
            self.gview(name, "bias").copy_(sums[:co])
        seg = self.segs[(name, "kernel")]
        dw_ptr = self.G.data_ptr() + 4 * seg.off
        dy_str = (dy.pix, dy.pix * dy.w, dy.pix * dy.w * dy.h)
        if k == 7:
            padded, src = self.saved[name]
            h, w = self.H, self.W
            self.wgrad(Act(padded, 4), dy.ptr, dy_str, dw_ptr, 32, co, 7, 1, 2, 1, 0, 0, h // 2, w // 2,
                       x_geom=(n, h + 6, w // 2, 8, (w + 8) * 4, (h + 6) * (w + 8) * 4), dy_t=dy.t)
            self._join_side()
            gk = self.gview(name, "kernel")
            srcl = self.g.layers[src]
            if srcl.cls == "BatchNormalization":
                # d beta[c] = sum_{co,tap} W[co][tap][c] * (sum of dY over the taps' valid pixels); the
                # indicator channel of the prepared input made the weight gradient hold those sums
                g5 = gk.view(-1, 7, 1, 8, 4)
                w5 = self.view(name, "kernel").view(-1, 7, 1, 8, 4).to(torch.bfloat16).float()  # as the forward used it
                dbeta = (w5[..., :3] * g5[..., 3:4]).sum(dim=(0, 1, 2, 3))
                self.gview(src, "beta").copy_(dbeta)
            gk.mul_(self.stem_mask[1])
            if self.grad_log is not None:
                self.grad_log[name] = (dy, None)
            return
        xin, pad = self.saved[name]
        cin = _c32(xin.c)
        self.wgrad(xin, dy.ptr, dy_str, dw_ptr, cin, (co + 7) // 8 * 8, k, k, stride, stride, pad, pad,
                   l.out.h, l.out.w, dy_t=dy.t)
        # data gradient
        src_name = self._src(l.inbound[0])[0]
        if self.g.layers[src_name].cls == "InputLayer":
            return
        ci = xin.c
        packs = self.dgrad_weight(l)
        dyv = Act(dy.t, cop, dy.c0)
        if stride == 1:
            narrow = dy.pix if dy.pix < cop else 0
            bn_mask = self._bn_mask_for(src_name, l.name, grads) if ci % 8 == 0 else None

            def run(res):
                out = self.new(xin.h, xin.w, ci)
                self._masked_dgrad(l.name, bn_mask if res is None else None, lambda bm: self.conv(
                    dyv, packs[0].data_ptr(), ci, k, k, 1, 1, k - 1 - pad, k - 1 - pad, out, xin.h, xin.w,
                    residual=res, in_valid_c=narrow, bn_mask=bm))
                return out
            self._fused_dgrad(grads, src_name, l.name, dy, run)
            return
        else:
            dx = self.new(xin.h, xin.w, ci, zero=True)
            for ph, pk in enumerate(packs):
                if pk is None:
                    continue
                wt, oy0, ox0, th, tw = pk
                a, b = ph >> 1, ph & 1
                hp, wp = (xin.h - a + 1) // 2, (xin.w - b + 1) // 2
                view = (dx.ptr + 2 * (a * dx.pix * dx.w + b * dx.pix), 2 * dx.pix, 2 * dx.pix * dx.w,
                        dx.pix * dx.w * dx.h)
                # out[y, x] = sum_t dY[y + oy0 + ty, x + ox0 + tx] * wt  ->  pad = -offset
                self.conv(dyv, wt.data_ptr(), ci, th, tw, 1, 1, -oy0, -ox0, None, hp, wp, out_view=view)
        self._accum(grads, src_name, dx, l.name, dy)

    def _convT_backward(self, l, dy, grads):
        name = l.name
        xin, _ = self.saved[name]
        co = l.out.c
        cin = _c32(xin.c)
        seg = self.segs[(name, "kernel")]
        per_phase = co * 4 * cin
        for a in range(2):
            for b in range(2):
                dyp = dy.ptr + 2 * (a * dy.pix * dy.w + b * dy.pix)
                self.wgrad(xin, dyp, (2 * dy.pix, 2 * dy.pix * dy.w, dy.pix * dy.w * dy.h),
                           self.G.data_ptr() + 4 * (seg.off + (a * 2 + b) * per_phase), cin, co, 2, 2, 1, 1,
                           1 - a, 1 - b, xin.h, xin.w, dy_t=dy.t)
        src_name = l.inbound[0]
        packs = self.dgrad_weight(l)
        wt = packs[0]
        dyv = Act(dy.t, _c32(co), dy.c0)
        # four phase lattices of dY accumulated in one launch of the ring kernel
        phased = (len(packs) > 1 and xin.c % 8 == 0 and dy.h * dy.pix * dy.w * 2 < 2 ** 31 and
                  not os.environ.get("MISEG_NO_PHASED_DGRAD"))

        bn_mask = self._bn_mask_for(src_name, l.name, grads) if xin.c % 8 == 0 else None

        def run(res):
            out = self.new(xin.h, xin.w, xin.c)
            if phased:
                launch = lambda bm: self.conv(dyv, packs[1].data_ptr(), xin.c, 2, 2, 1, 1, 0, 0, out, xin.h, xin.w,   # noqa: E731
                                              residual=res, transposed=2, bn_mask=bm)
            else:
                launch = lambda bm: self.conv(dyv, wt.data_ptr(), xin.c, 4, 4, 2, 2, 1, 1, out, xin.h, xin.w,   # noqa: E731
                                              residual=res, bn_mask=bm)
            self._masked_dgrad(l.name, bn_mask if res is None else None, launch)
            return out
        self._fused_dgrad(grads, src_name, l.name, dy, run)

    # ------------------------------------------------------------------ data parallel
    def _grads_final(self, name):
        """Layer `name`'s parameter gradients are written; everything stored after them in the flat
        buffer belongs to layers whose backward ran earlier."""
        if self.buckets is None:
            return
        lo = self.layer_lo.get(name)
        if lo is not None:
            if self.hi_pending(lo):
                self._join_side()             # the bucket about to leave includes side-stream gradients
            self.buckets.ready(lo)

    def hi_pending(self, lo):
        """Would announcing `lo` send a bucket now?"""
        b = self.buckets
        return b.hi - max(0, min(lo, b.hi)) >= b.bucket

    # ------------------------------------------------------------------ optimiser
    def adam_step(self):
        if self.buckets is not None:
            self.buckets.wait()
        self.step_count += 1
        t = self.step_count
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        L.check(self.lib.miseg_adam_step(self.P.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                         self.G.data_ptr(), self.PB.data_ptr(), self.nparam, lr_t, self.b1,
                                         self.b2, self.eps, 1.0, self.stream()))
        self._dgrad_cache = {}
        self.dirty = True                 # host copies of the weights are stale

    # ------------------------------------------------------------------ state
    def set_batch(self, n):
        """Images per step from now on (nothing persistent is sized by it)."""
        if n < 1:
            raise ValueError("batch must be positive")
        self.n = int(n)

    def state(self):
        return {"P": self.P, "M": self.M, "V": self.V, "step": self.step_count, "moving": self.moving}

    def load_state(self, st):
        self.P.copy_(st["P"])
        self.M.copy_(st["M"])
        self.V.copy_(st["V"])
        self.PB.copy_(self.P)
        self.step_count = st["step"]
        for k, (m, v) in st["moving"].items():
            self.moving[k][0].copy_(m)
            self.moving[k][1].copy_(v)
        self._dgrad_cache = {}

    def average_moving_stats(self):
        """BatchNormalization moving statistics are per replica; a mirrored strategy reads their
        mean over replicas when the model is saved or evaluated."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        flat = torch.cat([t for pair in self.moving.values() for t in pair])
        dist.all_reduce(flat)
        flat /= self.world
        off = 0
        for pair in self.moving.values():
            for t in pair:
                t.copy_(flat[off:off + t.numel()])
                off += t.numel()

    def profile_step(self, x, gt):
        """One training step with every kernel launch timed by HIP events on its own stream.
        Returns {kernel: {"launches", "ms", "flops", "bytes"}} (see profile.TimedLib)."""
        self.lib.records = []
        self.lib.on = True
        try:
            self.train_step(x, gt)
        finally:
            self.lib.on = False
        return self.lib.summary()

    def train_step(self, x, gt):
        """One optimisation step on a local batch.  Returns (sums, pixels) of the forward pass for
        loss / metric logging (segmentation_models sums, see losses.py)."""
        self.forward(x)
        sums, pixels = self.loss_and_metrics(gt)
        self.backward(loss_scale=1.0 / self.world)
        self.adam_step()
        return sums, pixels
