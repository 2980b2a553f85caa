"""Which stored tensor costs how much mask agreement?  TEST INFRASTRUCTURE: see oracle/__init__.py.

A mixed-precision run of the ResNet U-Net keeps, between kernels, every BatchNormalization(+ReLU)
output and every convolution kernel in a 16-bit format and everything else (accumulators, the residual
stream, logits) in fp32.  This module reruns the fp32 oracle (oracle/nets.py on oracle/ops_torch.py)
with the rounding switched on at a chosen subset of those sites, and reports, per subset, the share
of pixels whose strict-argmax mask (utils/utils.py:50-68 of the reference) differs from the
all-fp32 run's.  Site names: "W:<k>" = kernel of the k-th convolution call, "A:<k>" = output of the
k-th BatchNormalization call, both in graph order; `site_names()` maps them to layer names.

Formats: "bf16" (8 significand bits), "fp16" (11 bits), "bf16x2" (a value kept as the sum of two
bf16 numbers, hi + lo: 16 bits).

Run:  python -m oracle.attribution --size 960x1280 [--encoder resnet34] [--out file.json]
"""
import argparse
import json
import sys
import time

import numpy as np
import torch

from . import nets, post
from . import ops_torch as T


def _round(x, fmt):
    if fmt == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if fmt == "fp16":
        return x.to(torch.float16).to(torch.float32)
    if fmt == "bf16x2":
        hi = x.to(torch.bfloat16).to(torch.float32)
        return hi + (x - hi).to(torch.bfloat16).to(torch.float32)
    raise ValueError(fmt)


class SelectiveOps:
    """ops_torch with rounding at the sites in `sites` ({site: format}); `default` is the format
    of every site not listed (None = keep fp32)."""

    name = "torch-selective"

    def __init__(self, sites=None, default=None):
        self.sites = dict(sites or {})
        self.default = default
        self.nw = 0
        self.na = 0
        self.log = []

    def _fmt(self, site):
        return self.sites.get(site, self.default)

    def __getattr__(self, k):          # every op not overridden below
        return getattr(T, k)

    def _w(self, w):
        site = "W:%d" % self.nw
        self.nw += 1
        self.log.append(site)
        f = self._fmt(site)
        w = T._t(w)
        return _round(w, f) if f else w

    def conv2d(self, x, w, stride=1, pad=0, bias=None):
        return T.conv2d(x, self._w(w), stride=stride, pad=pad, bias=bias)

    def conv2d_transpose_4x4s2(self, x, w):
        return T.conv2d_transpose_4x4s2(x, self._w(w))

    def batchnorm(self, x, gamma, beta, mean, var, eps):
        site = "A:%d" % self.na
        self.na += 1
        self.log.append(site)
        y = T.batchnorm(x, gamma, beta, mean, var, eps)
        f = self._fmt(site)
        return _round(y, f) if f else y


def site_names(encoder, decoder, classes=3):
    """site -> Keras layer name, by replaying the graph on a tiny input."""
    spec = nets.build_spec(encoder, decoder, classes)
    w = nets.init_weights(spec, seed=1)
    calls = []

    class Rec(nets._Ctx):
        def bn(self, x, name, eps):
            calls.append(("A", name))
            return super().bn(x, name, eps)

        def conv(self, x, name, stride=1, pad=0):
            calls.append(("W", name))
            return super().conv(x, name, stride, pad)

        def convT(self, x, name):
            calls.append(("W", name))
            return super().convT(x, name)

    ctx_cls = nets._Ctx
    nets._Ctx = Rec
    try:
        nets.forward(T, encoder, decoder, w, np.zeros((1, 64, 64, 3), np.float32))
    finally:
        nets._Ctx = ctx_cls
    out, nw, na = {}, 0, 0
    for kind, name in calls:
        if kind == "W":
            out["W:%d" % nw] = name
            nw += 1
        else:
            out["A:%d" % na] = name
            na += 1
    return out


def masks(probs):
    return np.stack([post.collapse_probabilities(p.copy()) for p in probs])


class Study:
    def __init__(self, encoder, decoder, h, w, n=1, classes=3, seed=7, weights=None, x=None):
        self.enc, self.dec = encoder, decoder
        spec = nets.build_spec(encoder, decoder, classes)
        self.weights = weights or nets.init_weights(spec, seed=seed, randomize_bn=True)
        if x is None:
            rng = np.random.default_rng(1234)
            img = rng.integers(0, 256, (n, h, w, 3)).astype(np.uint8)
            yy, xx = np.mgrid[0:h, 0:w]
            img = (img * 0.5 + 64 * (1 + np.sin(xx / 7.0 + yy / 11.0))[None, :, :, None]).astype(np.uint8)
            x = np.stack([post.normalize_image(i) for i in img])
        self.x = x
        self.names = site_names(encoder, decoder, classes)
        with torch.no_grad():
            self.ref = T.to_numpy(nets.forward(T, encoder, decoder, self.weights, x))
        self.ref_mask = masks(self.ref)

    def run(self, sites=None, default=None):
        """-> (share of pixels whose mask differs from the fp32 run's, max |dp|)."""
        ops = SelectiveOps(sites, default)
        with torch.no_grad():
            got = T.to_numpy(nets.forward(ops, self.enc, self.dec, self.weights, self.x))
        differ = (masks(got) != self.ref_mask).any(axis=-1).mean()
        return float(differ), float(np.abs(got - self.ref).max())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", default="resnet34")
    ap.add_argument("--decoder", default="U-Net")
    ap.add_argument("--size", default="960x1280")
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args(argv)
    if a.threads:
        torch.set_num_threads(a.threads)
    h, w = (int(v) for v in a.size.split("x"))
    t0 = time.time()
    st = Study(a.encoder, a.decoder, h, w)
    sites = list(st.names)
    res = {"config": "%s %s %dx%d" % (a.encoder, a.decoder, h, w), "single": {}, "groups": {}}

    def rec(key, label, sites_=None, default=None):
        d, dp = st.run(sites_, default)
        res[key][label] = {"differ": d, "max_dp": dp}
        print("%-44s differ %.5f %%  max|dp| %.4f   (%.0f s)" % (label, 100 * d, dp, time.time() - t0), flush=True)
        return d

    rec("groups", "all bf16", default="bf16")
    rec("groups", "all fp16", default="fp16")
    rec("groups", "all bf16x2", default="bf16x2")
    rec("groups", "weights bf16 only", {s: "bf16" for s in sites if s[0] == "W"})
    rec("groups", "activations bf16 only", {s: "bf16" for s in sites if s[0] == "A"})
    single = {}
    for s in sites:
        single[s] = rec("single", "%s %s" % (s, st.names[s]), {s: "bf16"})
    # sites in the order of what each costs alone; lift them to fp32 one after the other
    order = sorted(sites, key=lambda s: -single[s])
    res["order"] = [(s, st.names[s], single[s]) for s in order]
    lifted = {}
    for k in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        if k > len(order):
            break
        lifted = {s: None for s in order[:k]}
        cfg = {s: "bf16" for s in sites if s not in lifted}
        rec("groups", "bf16 but the %d costliest sites in fp32" % k, cfg)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
