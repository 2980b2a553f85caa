"""Functional restatement of the network graphs that the reference builds at
utils/model.py:175-211 through segmentation_models 1.0.1 (`sm.Unet`,
`sm.Linknet`, `sm.FPN`) on the encoders of image-classifiers 1.0.0
(ResNet-18/34/50/101/152) and keras-applications 1.0.8 (DenseNet-121/169/201,
VGG-16/19).  TEST
INFRASTRUCTURE: see oracle/__init__.py.

The graph code is written once against an op back-end (`oracle.ops_np` or
`oracle.ops_torch`), so the two back-ends check each other's arithmetic while
the known-answer tests pin the graph (layer names, shapes, parameter counts).

Weights are a dict  layer_name -> {weight_name: float32 array}  keyed by the
Keras layer names an .h5 written by the reference would carry
(utils/model.py:265 loads by those names).
"""
from collections import OrderedDict

import numpy as np

RESNET_REPS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3),
               "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}
RESNET_BOTTLENECK = ("resnet50", "resnet101", "resnet152")      # [UPSTREAM] residual_bottleneck_block
DENSENET_BLOCKS = {"densenet169": (6, 12, 32, 32), "densenet121": (6, 12, 24, 16),
                   "densenet201": (6, 12, 48, 32)}
# keras-applications 1.0.8 vgg16.py / vgg19.py: 3x3 convolutions with bias and ReLU, five blocks of
# (64, 128, 256, 512, 512) filters, each closed by MaxPooling2D(2, strides=2); no batch normalisation
VGG_BLOCKS = {"vgg16": (2, 2, 3, 3, 3), "vgg19": (2, 2, 4, 4, 4)}
VGG_FILTERS = (64, 128, 256, 512, 512)
RESNET_EPS = 2e-5
DENSENET_EPS = 1.001e-5
DECODER_EPS = 1e-3
UNET_FILTERS = (512, 256, 128, 64, 32)          # utils/model.py:184
LINKNET_LAST_FILTERS = 16                       # utils/model.py:195
FPN_PYRAMID = 256                               # utils/model.py:207
FPN_SEGMENTATION = 128                          # sm.FPN default


# ----------------------------------------------------------------------------
# parameter specification (names, shapes, initialisers) in layer-creation order
# ----------------------------------------------------------------------------
class Spec:
    def __init__(self):
        self.layers = OrderedDict()

    def conv(self, name, k, cin, cout, bias=False, init="he_uniform"):
        ws = [("kernel", (k, k, cin, cout), init)]
        if bias:
            ws.append(("bias", (cout,), "zeros"))
        self.layers[name] = ("conv", ws)

    def convT(self, name, cin, cout):
        # Keras stores Conv2DTranspose kernels as (kh, kw, Cout, Cin)
        self.layers[name] = ("convT", [("kernel", (4, 4, cout, cin), "glorot_uniform")])

    def bn(self, name, c, scale=True):
        ws = []
        if scale:
            ws.append(("gamma", (c,), "ones"))
        ws += [("beta", (c,), "zeros"), ("moving_mean", (c,), "zeros"),
               ("moving_variance", (c,), "ones")]
        self.layers[name] = ("bn", ws)


def _resnet_spec(s, reps, bottleneck=False):
    s.bn("bn_data", 3, scale=False)
    s.conv("conv0", 7, 3, 64)
    s.bn("bn0", 64)
    cin = 64
    ex = 4 if bottleneck else 1
    for stage, rep in enumerate(reps):
        f = 64 * 2 ** stage
        for block in range(rep):
            base = "stage%d_unit%d_" % (stage + 1, block + 1)
            s.bn(base + "bn1", cin)
            if block == 0:
                s.conv(base + "sc", 1, cin, f * ex)
            if bottleneck:
                s.conv(base + "conv1", 1, cin, f)
                s.bn(base + "bn2", f)
                s.conv(base + "conv2", 3, f, f)
                s.bn(base + "bn3", f)
                s.conv(base + "conv3", 1, f, 4 * f)
            else:
                s.conv(base + "conv1", 3, cin, f)
                s.bn(base + "bn2", f)
                s.conv(base + "conv2", 3, f, f)
            cin = f * ex
    s.bn("bn1", cin)
    return [256 * ex, 128 * ex, 64 * ex, 64], cin     # skip channels (deep -> shallow), output channels


def _densenet_spec(s, blocks):
    s.conv("conv1/conv", 7, 3, 64, init="glorot_uniform")
    s.bn("conv1/bn", 64)
    c = 64
    skip_ch = [64]
    for bi, nl in enumerate(blocks):
        name = "conv%d" % (bi + 2)
        for i in range(nl):
            b = "%s_block%d" % (name, i + 1)
            s.bn(b + "_0_bn", c)
            s.conv(b + "_1_conv", 1, c, 128, init="glorot_uniform")
            s.bn(b + "_1_bn", 128)
            s.conv(b + "_2_conv", 3, 128, 32, init="glorot_uniform")
            c += 32
        if bi < len(blocks) - 1:
            p = "pool%d" % (bi + 2)
            s.bn(p + "_bn", c)
            s.conv(p + "_conv", 1, c, c // 2, init="glorot_uniform")
            c //= 2
            skip_ch.append(c)
    s.bn("bn", c)
    return skip_ch[::-1], c


def _vgg_spec(s, blocks):
    cin = 3
    for b, (reps, f) in enumerate(zip(blocks, VGG_FILTERS)):
        for i in range(reps):
            s.conv("block%d_conv%d" % (b + 1, i + 1), 3, cin, f, bias=True, init="glorot_uniform")
            cin = f
    # sm Backbones feature layers: the last convolution of every block, deep -> shallow; the
    # decoders take get_feature_layers(name, n=4), i.e. the four deepest
    return list(VGG_FILTERS[::-1])[:4], cin


def _conv_bn(s, name, k, cin, cout):
    s.conv(name + "_conv", k, cin, cout)
    s.bn(name + "_bn", cout)


def build_spec(encoder, decoder, classes):
    """Parameter specification of make_model(encoder, decoder, ..., classes)
    (utils/model.py:145-236)."""
    s = Spec()
    if encoder in RESNET_REPS:
        skips, c = _resnet_spec(s, RESNET_REPS[encoder], encoder in RESNET_BOTTLENECK)
    elif encoder in DENSENET_BLOCKS:
        skips, c = _densenet_spec(s, DENSENET_BLOCKS[encoder])
    elif encoder in VGG_BLOCKS:
        skips, c = _vgg_spec(s, VGG_BLOCKS[encoder])
        if decoder in ("U-Net", "Linknet"):
            # [UPSTREAM] build_unet / build_linknet: a backbone that ends in MaxPooling2D gets two
            # Conv3x3BnReLU(512) "center" blocks in front of the decoder
            _conv_bn(s, "center_block1", 3, c, 512)
            _conv_bn(s, "center_block2", 3, 512, 512)
            c = 512
    else:
        raise ValueError("unsupported encoder %r" % (encoder,))
    if decoder == "U-Net":
        for i, f in enumerate(UNET_FILTERS):
            s.convT("decoder_stage%da_transpose" % i, c, f)
            s.bn("decoder_stage%da_bn" % i, f)
            cc = f + (skips[i] if i < len(skips) else 0)
            _conv_bn(s, "decoder_stage%db" % i, 3, cc, f)
            c = f
        s.conv("final_conv", 3, c, classes, bias=True, init="glorot_uniform")
    elif decoder == "Linknet":
        for i in range(5):
            out = skips[i] if i < len(skips) else LINKNET_LAST_FILTERS
            _conv_bn(s, "decoder_stage%da" % i, 1, c, c // 4)
            s.convT("decoder_stage%db_transpose" % i, c // 4, c // 4)
            s.bn("decoder_stage%db_bn" % i, c // 4)
            _conv_bn(s, "decoder_stage%dc" % i, 1, c // 4, out)
            c = out
        s.conv("final_conv", 3, c, classes, bias=True, init="glorot_uniform")
    elif decoder == "FPN":
        for j, st in enumerate((5, 4, 3, 2)):
            if c != FPN_PYRAMID:
                s.conv("fpn_stage_p%d_pre_conv" % st, 1, c, FPN_PYRAMID, bias=True)
            s.conv("fpn_stage_p%d_conv" % st, 1, skips[j], FPN_PYRAMID, bias=True)
            c = FPN_PYRAMID
        for st in (5, 4, 3, 2):
            _conv_bn(s, "segm_stage%da" % st, 3, FPN_PYRAMID, FPN_SEGMENTATION)
            _conv_bn(s, "segm_stage%db" % st, 3, FPN_SEGMENTATION, FPN_SEGMENTATION)
        _conv_bn(s, "final_stage", 3, 4 * FPN_SEGMENTATION, FPN_SEGMENTATION)
        s.conv("head_conv", 3, FPN_SEGMENTATION, classes, bias=True, init="glorot_uniform")
    else:
        raise ValueError("unsupported decoder %r" % (decoder,))
    return s.layers


def count_params(spec):
    return sum(int(np.prod(shape)) for _, ws in spec.values() for _, shape, _ in ws)


def init_weights(spec, seed=1234, randomize_bn=False):
    """Keras initialisers (he_uniform / glorot_uniform / zeros / ones) from a
    seeded numpy generator.  `randomize_bn` draws non-trivial BatchNorm
    parameters and moving statistics so inference-mode BN is not an identity."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for lname, (kind, ws) in spec.items():
        d = OrderedDict()
        for wname, shape, init in ws:
            if init in ("he_uniform", "glorot_uniform"):
                rf = shape[0] * shape[1]
                fan_a, fan_b = shape[2] * rf, shape[3] * rf
                lim = np.sqrt(6.0 / fan_a) if init == "he_uniform" else np.sqrt(6.0 / (fan_a + fan_b))
                d[wname] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif init == "zeros":
                d[wname] = np.zeros(shape, np.float32)
            elif init == "ones":
                d[wname] = np.ones(shape, np.float32)
            else:
                raise ValueError(init)
        if randomize_bn:
            if kind == "bn":
                c = d["beta"].shape[0]
                if "gamma" in d:
                    d["gamma"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
                d["beta"] = rng.normal(0, 0.1, c).astype(np.float32)
                d["moving_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
                d["moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            elif "bias" in d:
                d["bias"] = rng.normal(0, 0.1, d["bias"].shape).astype(np.float32)
        out[lname] = d
    return out


# ----------------------------------------------------------------------------
# forward graphs
# ----------------------------------------------------------------------------
class _Ctx:
    def __init__(self, ops, weights, taps, training=False):
        self.ops, self.w, self.taps, self.training = ops, weights, taps, training
        self.batch_stats = OrderedDict()      # training mode: layer -> (mean, biased variance)

    def bn(self, x, name, eps):
        p = self.w[name]
        if self.training:
            # Keras BatchNormalization under model.fit (model_train.py:284): normalise with the
            # statistics of the batch; the moving averages are updated from them on the side
            y, mean, var = self.ops.batchnorm_train(x, p.get("gamma"), p["beta"], eps)
            self.batch_stats[name] = (mean, var)
            return y
        return self.ops.batchnorm(x, p.get("gamma"), p["beta"], p["moving_mean"],
                                  p["moving_variance"], eps)

    def conv(self, x, name, stride=1, pad=0):
        p = self.w[name]
        return self.ops.conv2d(x, p["kernel"], stride=stride, pad=pad, bias=p.get("bias"))

    def convT(self, x, name):
        return self.ops.conv2d_transpose_4x4s2(x, self.w[name]["kernel"])

    def tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = self.ops.to_numpy(x)
        return x

    def conv_bn_relu(self, x, name, k):
        x = self.conv(x, name + "_conv", pad=(k - 1) // 2)
        x = self.bn(x, name + "_bn", DECODER_EPS)
        return self.tap(name + "_relu", self.ops.relu(x))


def _resnet_forward(c, x, reps, bottleneck=False):
    ops = c.ops
    x = c.bn(x, "bn_data", RESNET_EPS)
    x = c.conv(x, "conv0", stride=2, pad=3)
    x = ops.relu(c.bn(x, "bn0", RESNET_EPS))
    skips = {"relu0": c.tap("relu0", x)}
    x = ops.maxpool3x3s2_pad1(x)
    for stage, rep in enumerate(reps):
        for block in range(rep):
            base = "stage%d_unit%d_" % (stage + 1, block + 1)
            a = ops.relu(c.bn(x, base + "bn1", RESNET_EPS))
            c.tap(base + "relu1", a)
            if block == 0:
                stride = 1 if stage == 0 else 2
                shortcut = c.conv(a, base + "sc", stride=stride, pad=0)
                skips[base + "relu1"] = a
            else:
                stride = 1
                shortcut = x
            if bottleneck:      # 1x1 -> 3x3 (carries the stride) -> 1x1 to four times the filters
                y = c.conv(a, base + "conv1", stride=1, pad=0)
                y = ops.relu(c.bn(y, base + "bn2", RESNET_EPS))
                y = c.conv(y, base + "conv2", stride=stride, pad=1)
                y = ops.relu(c.bn(y, base + "bn3", RESNET_EPS))
                y = c.conv(y, base + "conv3", stride=1, pad=0)
            else:
                y = c.conv(a, base + "conv1", stride=stride, pad=1)
                y = ops.relu(c.bn(y, base + "bn2", RESNET_EPS))
                y = c.conv(y, base + "conv2", stride=1, pad=1)
            x = c.tap(base + "add", ops.add(y, shortcut))
    x = ops.relu(c.bn(x, "bn1", RESNET_EPS))
    c.tap("relu1", x)
    # sm Backbones feature layers for the ResNets, deep -> shallow
    feats = [skips["stage4_unit1_relu1"], skips["stage3_unit1_relu1"],
             skips["stage2_unit1_relu1"], skips["relu0"]]
    return x, feats


def _densenet_forward(c, x, blocks):
    ops = c.ops
    x = c.conv(x, "conv1/conv", stride=2, pad=3)
    x = ops.relu(c.bn(x, "conv1/bn", DENSENET_EPS))
    feats = [c.tap("conv1/relu", x)]
    x = ops.maxpool3x3s2_pad1(x)
    for bi, nl in enumerate(blocks):
        name = "conv%d" % (bi + 2)
        for i in range(nl):
            b = "%s_block%d" % (name, i + 1)
            y = ops.relu(c.bn(x, b + "_0_bn", DENSENET_EPS))
            y = c.conv(y, b + "_1_conv")
            y = ops.relu(c.bn(y, b + "_1_bn", DENSENET_EPS))
            y = c.conv(y, b + "_2_conv", pad=1)
            x = ops.concat([x, y])
        c.tap(name + "_out", x)
        if bi < len(blocks) - 1:
            p = "pool%d" % (bi + 2)
            y = ops.relu(c.bn(x, p + "_bn", DENSENET_EPS))
            y = c.tap(p + "_conv", c.conv(y, p + "_conv"))
            feats.append(y)                      # skip = transition conv output, before pooling
            x = ops.avgpool2x2(y)
    x = ops.relu(c.bn(x, "bn", DENSENET_EPS))
    c.tap("relu", x)
    return x, feats[::-1]


def _vgg_forward(c, x, blocks):
    ops = c.ops
    feats = []
    for b, reps in enumerate(blocks):
        for i in range(reps):
            name = "block%d_conv%d" % (b + 1, i + 1)
            x = c.tap(name, ops.relu(c.conv(x, name, pad=1)))
        feats.append(x)
        x = ops.maxpool2x2(x)
    return x, feats[::-1][:4]            # get_feature_layers(name, n=4): block1_conv2 is not a skip


def _center(c, x):
    """Two Conv3x3BnReLU(512) blocks when the backbone ends in a pooling layer (the VGGs)."""
    if "center_block1_conv" in c.w:
        x = c.conv_bn_relu(x, "center_block1", 3)
        x = c.conv_bn_relu(x, "center_block2", 3)
    return x


def _unet_decoder(c, x, feats):
    ops = c.ops
    x = _center(c, x)
    for i in range(5):
        x = c.convT(x, "decoder_stage%da_transpose" % i)
        x = ops.relu(c.bn(x, "decoder_stage%da_bn" % i, DECODER_EPS))
        if i < len(feats):
            x = ops.concat([x, feats[i]])
        x = c.conv_bn_relu(x, "decoder_stage%db" % i, 3)
    return c.conv(x, "final_conv", pad=1)


def _linknet_decoder(c, x, feats):
    ops = c.ops
    x = _center(c, x)
    for i in range(5):
        x = c.conv_bn_relu(x, "decoder_stage%da" % i, 1)
        x = c.convT(x, "decoder_stage%db_transpose" % i)
        x = ops.relu(c.bn(x, "decoder_stage%db_bn" % i, DECODER_EPS))
        x = c.conv_bn_relu(x, "decoder_stage%dc" % i, 1)
        if i < len(feats):
            x = ops.add(x, feats[i])
        c.tap("decoder_stage%d_out" % i, x)
    return c.conv(x, "final_conv", pad=1)


def _fpn_decoder(c, x, feats):
    ops = c.ops
    pyr = []
    for j, st in enumerate((5, 4, 3, 2)):
        pre = "fpn_stage_p%d_pre_conv" % st
        if pre in c.w:
            x = c.conv(x, pre)
        skip = c.conv(feats[j], "fpn_stage_p%d_conv" % st)
        x = ops.add(ops.upsample_nearest(x, 2), skip)
        pyr.append(c.tap("fpn_stage_p%d_add" % st, x))
    segs = []
    for p, st, up in zip(pyr, (5, 4, 3, 2), (8, 4, 2, 1)):
        s = c.conv_bn_relu(p, "segm_stage%da" % st, 3)
        s = c.conv_bn_relu(s, "segm_stage%db" % st, 3)
        if up > 1:
            s = ops.upsample_nearest(s, up)
        segs.append(s)
    x = ops.concat([segs[3], segs[2], segs[1], segs[0]])   # [s2, s3, s4, s5]
    x = c.conv_bn_relu(x, "final_stage", 3)
    x = ops.upsample_bilinear2x(x)
    return c.conv(x, "head_conv", pad=1)


def forward(ops, encoder, decoder, weights, x, activation="softmax", temperature=1.0,
            taps=None, return_logits=False, training=False, batch_stats=None):
    """Forward pass: inference mode is model(batch, training=False) at
    utils/predict.py:159; `training=True` is the forward half of a Model.fit step
    (model_train.py:284), BatchNormalization on batch statistics, which are returned in
    the dict `batch_stats` when one is passed.  `x`: (N,H,W,3) float32 in [0,1].  Returns
    probabilities (N,H,W,classes) as a back-end tensor (and logits if asked)."""
    c = _Ctx(ops, {k: {n: ops.asarray(v) for n, v in d.items()} for k, d in weights.items()}, taps,
             training)
    if batch_stats is not None:
        c.batch_stats = batch_stats
    x = ops.asarray(x)
    if encoder in RESNET_REPS:
        x, feats = _resnet_forward(c, x, RESNET_REPS[encoder], encoder in RESNET_BOTTLENECK)
    elif encoder in VGG_BLOCKS:
        x, feats = _vgg_forward(c, x, VGG_BLOCKS[encoder])
    else:
        x, feats = _densenet_forward(c, x, DENSENET_BLOCKS[encoder])
    if decoder == "U-Net":
        logits = _unet_decoder(c, x, feats)
    elif decoder == "Linknet":
        logits = _linknet_decoder(c, x, feats)
    elif decoder == "FPN":
        logits = _fpn_decoder(c, x, feats)
    else:
        raise ValueError(decoder)
    c.tap("logits", logits)
    if activation == "softmax":
        probs = ops.softmax(logits, temperature)
    elif activation == "sigmoid":
        probs = ops.sigmoid(logits)
    else:
        raise ValueError(activation)
    return (probs, logits) if return_logits else probs


def densenet_layer_names(blocks=(6, 12, 32, 32)):
    """Keras `model.layers` names of keras_applications DenseNet (no top), in
    index order; segmentation_models picks DenseNet-169 skips by the indices
    (367, 139, 51, 4)."""
    names = ["input_1", "zero_padding2d", "conv1/conv", "conv1/bn", "conv1/relu",
             "zero_padding2d_1", "pool1"]
    for bi, nl in enumerate(blocks):
        name = "conv%d" % (bi + 2)
        for i in range(nl):
            b = "%s_block%d" % (name, i + 1)
            names += [b + s for s in ("_0_bn", "_0_relu", "_1_conv", "_1_bn", "_1_relu",
                                      "_2_conv", "_concat")]
        if bi < len(blocks) - 1:
            p = "pool%d" % (bi + 2)
            names += [p + "_bn", p + "_relu", p + "_conv", p + "_pool"]
    names += ["bn", "relu"]
    return names
