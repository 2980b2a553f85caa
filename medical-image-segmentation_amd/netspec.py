"""Symbolic layer graphs of the networks `make_model` builds
(/root/reference utils/model.py:175-211): encoders ResNet-18/34/50/101/152
(classification_models), DenseNet-121/169/201 and VGG-16/19 (keras_applications), decoders
U-Net, Linknet and FPN (segmentation_models) with the hyper-parameters the
reference fixes.  A graph is an ordered list of Keras-style layer records
(name, class, config, inbound layers, weights); it carries the layer and weight
names an .h5 written by the reference uses, drives weight initialisation,
`summary()`, `to_json()` and is what the engine lowers to kernel launches.
"""
from collections import OrderedDict

BACKBONE_FEATURES = {
    # segmentation_models Backbones._default_feature_layers (deep -> shallow)
    "resnet18": ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0"),
    "resnet34": ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0"),
    "resnet50": ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0"),
    "resnet101": ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0"),
    "resnet152": ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0"),
    "vgg16": ("block5_conv3", "block4_conv3", "block3_conv3", "block2_conv2", "block1_conv2"),
    "vgg19": ("block5_conv4", "block4_conv4", "block3_conv4", "block2_conv2", "block1_conv2"),
    "densenet121": (311, 139, 51, 4),
    "densenet169": (367, 139, 51, 4),
    "densenet201": (479, 139, 51, 4),
}
RESNET_UNITS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3),
                "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}
# classification_models residual_bottleneck_block (1x1, 3x3, 1x1 to four times the filters) instead of
# residual_conv_block (3x3, 3x3)
RESNET_BOTTLENECK = ("resnet50", "resnet101", "resnet152")
VGG_BLOCKS = {"vgg16": (2, 2, 3, 3, 3), "vgg19": (2, 2, 4, 4, 4)}      # keras_applications vgg16.py / vgg19.py
DENSENET_LAYERS = {"densenet121": (6, 12, 24, 16), "densenet169": (6, 12, 32, 32),
                   "densenet201": (6, 12, 48, 32)}


class Sym:
    """A symbolic NHWC tensor: producing layer name and (H, W, C)."""
    __slots__ = ("layer", "h", "w", "c")

    def __init__(self, layer, h, w, c):
        self.layer, self.h, self.w, self.c = layer, h, w, c


class Layer:
    __slots__ = ("name", "cls", "config", "inbound", "weights", "out")

    def __init__(self, name, cls, config, inbound, weights, out):
        self.name, self.cls, self.config = name, cls, config
        self.inbound, self.weights, self.out = inbound, weights, out

    @property
    def output_shape(self):
        return (None, self.out.h, self.out.w, self.out.c)

    def count_params(self):
        n = 0
        for _, shape, _ in self.weights:
            k = 1
            for s in shape:
                k *= s
            n += k
        return n


class Graph:
    def __init__(self, name):
        self.name = name
        self.layers = OrderedDict()
        self._auto = {}

    # -- helpers ------------------------------------------------------------
    def _name(self, base, name):
        if name:
            return name
        k = self._auto.get(base, 0)
        self._auto[base] = k + 1
        return base if k == 0 else "%s_%d" % (base, k)

    def _add(self, name, cls, config, inputs, weights, h, w, c):
        if name in self.layers:
            raise ValueError("duplicate layer name %r" % name)
        out = Sym(name, h, w, c)
        self.layers[name] = Layer(name, cls, config, [s.layer for s in inputs], weights, out)
        return out

    # -- layer constructors ---------------------------------------------------
    def input(self, h, w, c, name="data"):
        return self._add(name, "InputLayer", {"batch_input_shape": [None, h, w, c]}, [], [], h, w, c)

    def zero_pad(self, x, p, name=None):
        return self._add(self._name("zero_padding2d", name), "ZeroPadding2D", {"padding": p}, [x], [],
                         x.h + 2 * p, x.w + 2 * p, x.c)

    def conv(self, x, filters, k, name, strides=1, padding="valid", use_bias=False,
             init="glorot_uniform", activation="linear"):
        if padding == "same":
            h, w = -(-x.h // strides), -(-x.w // strides)
        else:
            h, w = (x.h - k) // strides + 1, (x.w - k) // strides + 1
        ws = [("kernel", (k, k, x.c, filters), init)]
        if use_bias:
            ws.append(("bias", (filters,), "zeros"))
        cfg = {"filters": filters, "kernel_size": k, "strides": strides, "padding": padding,
               "use_bias": use_bias, "kernel_initializer": init, "activation": activation}
        return self._add(name, "Conv2D", cfg, [x], ws, h, w, filters)

    def conv_transpose(self, x, filters, name):
        ws = [("kernel", (4, 4, filters, x.c), "glorot_uniform")]
        cfg = {"filters": filters, "kernel_size": 4, "strides": 2, "padding": "same",
               "use_bias": False, "kernel_initializer": "glorot_uniform"}
        return self._add(name, "Conv2DTranspose", cfg, [x], ws, 2 * x.h, 2 * x.w, filters)

    def bn(self, x, name, epsilon, scale=True, momentum=0.99):
        ws = ([("gamma", (x.c,), "ones")] if scale else []) + [
            ("beta", (x.c,), "zeros"), ("moving_mean", (x.c,), "zeros"),
            ("moving_variance", (x.c,), "ones")]
        cfg = {"axis": 3, "epsilon": epsilon, "momentum": momentum, "scale": scale, "center": True}
        return self._add(name, "BatchNormalization", cfg, [x], ws, x.h, x.w, x.c)

    def act(self, x, kind, name):
        return self._add(name, "Activation", {"activation": kind}, [x], [], x.h, x.w, x.c)

    def tss(self, x, temperature, name="temperature_scaled_softmax"):
        return self._add(name, "TemperatureScaledSoftmax", {"temperature": temperature}, [x], [],
                         x.h, x.w, x.c)

    def maxpool(self, x, name, pool=3):
        return self._add(name, "MaxPooling2D", {"pool_size": pool, "strides": 2, "padding": "valid"},
                         [x], [], (x.h - pool) // 2 + 1, (x.w - pool) // 2 + 1, x.c)

    def avgpool(self, x, name):
        return self._add(name, "AveragePooling2D", {"pool_size": 2, "strides": 2}, [x], [],
                         x.h // 2, x.w // 2, x.c)

    def add(self, xs, name=None):
        return self._add(self._name("add", name), "Add", {}, xs, [], xs[0].h, xs[0].w, xs[0].c)

    def concat(self, xs, name):
        return self._add(name, "Concatenate", {"axis": 3}, xs, [], xs[0].h, xs[0].w,
                         sum(s.c for s in xs))

    def upsample(self, x, size, name, interpolation="nearest"):
        return self._add(name, "UpSampling2D", {"size": size, "interpolation": interpolation}, [x],
                         [], x.h * size, x.w * size, x.c)

    # -- queries ----------------------------------------------------------------
    def sym(self, key):
        """Layer output by name, or by Keras layer index."""
        if isinstance(key, int):
            return list(self.layers.values())[key].out
        return self.layers[key].out

    def count_params(self):
        return sum(l.count_params() for l in self.layers.values())

    def weight_spec(self):
        """[(layer_name, weight_name, shape, initializer)] in get_weights() order."""
        return [(l.name, wn, shape, init) for l in self.layers.values()
                for wn, shape, init in l.weights]


# ---------------------------------------------------------------------------
# encoders
# ---------------------------------------------------------------------------
def _resnet(g, x, units, bottleneck=False):
    eps = 2e-5
    he = "he_uniform"
    x = g.bn(x, "bn_data", eps, scale=False)
    x = g.zero_pad(x, 3)
    x = g.conv(x, 64, 7, "conv0", strides=2, init=he)
    x = g.bn(x, "bn0", eps)
    x = g.act(x, "relu", "relu0")
    x = g.zero_pad(x, 1)
    x = g.maxpool(x, "pooling0")
    for stage, reps in enumerate(units):
        filters = 64 * (2 ** stage)
        for block in range(reps):
            base = "stage%d_unit%d_" % (stage + 1, block + 1)
            first = block == 0
            strides = 2 if (first and stage > 0) else 1
            y = g.bn(x, base + "bn1", eps)
            y = g.act(y, "relu", base + "relu1")
            if bottleneck:
                shortcut = g.conv(y, 4 * filters, 1, base + "sc", strides=strides, init=he) if first else x
                y = g.conv(y, filters, 1, base + "conv1", init=he)
                y = g.bn(y, base + "bn2", eps)
                y = g.act(y, "relu", base + "relu2")
                y = g.zero_pad(y, 1)
                y = g.conv(y, filters, 3, base + "conv2", strides=strides, init=he)
                y = g.bn(y, base + "bn3", eps)
                y = g.act(y, "relu", base + "relu3")
                y = g.conv(y, 4 * filters, 1, base + "conv3", init=he)
            else:
                shortcut = g.conv(y, filters, 1, base + "sc", strides=strides, init=he) if first else x
                y = g.zero_pad(y, 1)
                y = g.conv(y, filters, 3, base + "conv1", strides=strides, init=he)
                y = g.bn(y, base + "bn2", eps)
                y = g.act(y, "relu", base + "relu2")
                y = g.zero_pad(y, 1)
                y = g.conv(y, filters, 3, base + "conv2", init=he)
            x = g.add([y, shortcut])
    x = g.bn(x, "bn1", eps)
    return g.act(x, "relu", "relu1")


def _vgg(g, x, blocks):
    for b, (reps, filters) in enumerate(zip(blocks, (64, 128, 256, 512, 512))):
        for i in range(reps):
            x = g.conv(x, filters, 3, "block%d_conv%d" % (b + 1, i + 1), padding="same", use_bias=True,
                       activation="relu")
        x = g.maxpool(x, "block%d_pool" % (b + 1), pool=2)
    return x


def _densenet(g, x, block_layers):
    eps = 1.001e-5
    x = g.zero_pad(x, 3)
    x = g.conv(x, 64, 7, "conv1/conv", strides=2)
    x = g.bn(x, "conv1/bn", eps)
    x = g.act(x, "relu", "conv1/relu")
    x = g.zero_pad(x, 1)
    x = g.maxpool(x, "pool1")
    for bi, nlayers in enumerate(block_layers):
        blk = "conv%d" % (bi + 2)
        for li in range(nlayers):
            n = "%s_block%d" % (blk, li + 1)
            y = g.bn(x, n + "_0_bn", eps)
            y = g.act(y, "relu", n + "_0_relu")
            y = g.conv(y, 128, 1, n + "_1_conv")
            y = g.bn(y, n + "_1_bn", eps)
            y = g.act(y, "relu", n + "_1_relu")
            y = g.conv(y, 32, 3, n + "_2_conv", padding="same")
            x = g.concat([x, y], n + "_concat")
        if bi + 1 < len(block_layers):
            t = "pool%d" % (bi + 2)
            y = g.bn(x, t + "_bn", eps)
            y = g.act(y, "relu", t + "_relu")
            y = g.conv(y, x.c // 2, 1, t + "_conv")
            x = g.avgpool(y, t + "_pool")
    x = g.bn(x, "bn", eps)
    return g.act(x, "relu", "relu")


# ---------------------------------------------------------------------------
# decoders (segmentation_models); Keras-default BN epsilon 1e-3
# ---------------------------------------------------------------------------
DEC_EPS = 1e-3


def _conv_bn_relu(g, x, filters, k, name):
    x = g.conv(x, filters, k, name + "_conv", padding="same", init="he_uniform")
    x = g.bn(x, name + "_bn", DEC_EPS)
    return g.act(x, "relu", name + "_relu")


def _center(g, x):
    """segmentation_models build_unet / build_linknet: a backbone ending in MaxPooling2D (the VGGs)
    gets two Conv3x3BnReLU(512) blocks in front of the decoder."""
    if g.layers[x.layer].cls == "MaxPooling2D":
        x = _conv_bn_relu(g, x, 512, 3, "center_block1")
        x = _conv_bn_relu(g, x, 512, 3, "center_block2")
    return x


def _unet(g, x, skips, classes):
    x = _center(g, x)
    for i, f in enumerate((512, 256, 128, 64, 32)):
        x = g.conv_transpose(x, f, "decoder_stage%da_transpose" % i)
        x = g.bn(x, "decoder_stage%da_bn" % i, DEC_EPS)
        x = g.act(x, "relu", "decoder_stage%da_relu" % i)
        if i < len(skips):
            x = g.concat([x, skips[i]], "decoder_stage%d_concat" % i)
        x = _conv_bn_relu(g, x, f, 3, "decoder_stage%db" % i)
    return g.conv(x, classes, 3, "final_conv", padding="same", use_bias=True)


def _linknet(g, x, skips, classes):
    x = _center(g, x)
    for i in range(5):
        cin = x.c
        cout = skips[i].c if i < len(skips) else 16
        x = _conv_bn_relu(g, x, cin // 4, 1, "decoder_stage%da" % i)
        x = g.conv_transpose(x, cin // 4, "decoder_stage%db_transpose" % i)
        x = g.bn(x, "decoder_stage%db_bn" % i, DEC_EPS)
        x = g.act(x, "relu", "decoder_stage%db_relu" % i)
        x = _conv_bn_relu(g, x, cout, 1, "decoder_stage%dc" % i)
        if i < len(skips):
            x = g.add([x, skips[i]], "decoder_stage%d_add" % i)
    return g.conv(x, classes, 3, "final_conv", padding="same", use_bias=True)


def _fpn(g, x, skips, classes, pyramid=256, segmentation=128):
    levels = []
    for skip, stage in zip(skips, (5, 4, 3, 2)):
        if x.c != pyramid:
            x = g.conv(x, pyramid, 1, "fpn_stage_p%d_pre_conv" % stage, use_bias=True,
                       init="he_uniform")
        lateral = g.conv(skip, pyramid, 1, "fpn_stage_p%d_conv" % stage, use_bias=True,
                         init="he_uniform")
        up = g.upsample(x, 2, "fpn_stage_p%d_upsampling" % stage)
        x = g.add([up, lateral], "fpn_stage_p%d_add" % stage)
        levels.append(x)
    heads = []
    for p, stage in zip(levels, (5, 4, 3, 2)):
        s = _conv_bn_relu(g, p, segmentation, 3, "segm_stage%da" % stage)
        heads.append(_conv_bn_relu(g, s, segmentation, 3, "segm_stage%db" % stage))
    s5 = g.upsample(heads[0], 8, "upsampling_stage5")
    s4 = g.upsample(heads[1], 4, "upsampling_stage4")
    s3 = g.upsample(heads[2], 2, "upsampling_stage3")
    x = g.concat([heads[3], s3, s4, s5], "aggregation_concat")
    x = _conv_bn_relu(g, x, segmentation, 3, "final_stage")
    x = g.upsample(x, 2, "final_upsampling", interpolation="bilinear")
    return g.conv(x, classes, 3, "head_conv", padding="same", use_bias=True)


def build_graph(encoder, decoder, input_shape, classes, activation=None, temperature=None,
                name=None):
    """Graph of make_model(encoder, decoder, input_shape, classes, ...).
    `temperature` set: the final activation is TemperatureScaledSoftmax (use_tss)."""
    h, w, c = input_shape
    if c != 3:
        raise ValueError("the encoders take 3-channel images, got input_shape=%r" % (input_shape,))
    if h % 32 or w % 32:
        raise ValueError("height and width must be multiples of 32, got %dx%d" % (h, w))
    if decoder not in ("U-Net", "Linknet", "FPN"):
        # utils/model.py:175-225 has no else branch: an unknown decoder leaves `model` unbound
        raise UnboundLocalError("local variable 'model' referenced before assignment "
                                "(unknown decoder %r)" % (decoder,))
    g = Graph(name or "model")
    x = g.input(h, w, c, "data" if encoder in RESNET_UNITS else "input_1")
    if encoder in RESNET_UNITS:
        x = _resnet(g, x, RESNET_UNITS[encoder], encoder in RESNET_BOTTLENECK)
    elif encoder in VGG_BLOCKS:
        x = _vgg(g, x, VGG_BLOCKS[encoder])
    elif encoder in DENSENET_LAYERS:
        x = _densenet(g, x, DENSENET_LAYERS[encoder])
    else:
        raise ValueError("unsupported encoder %r (supported: %s)" %
                         (encoder, sorted(list(RESNET_UNITS) + list(DENSENET_LAYERS) + list(VGG_BLOCKS))))
    # segmentation_models' Unet, Linknet and FPN all ask Backbones.get_feature_layers(name, n=4): of the VGGs'
    # five listed layers the four deepest are skips and the last decoder stage has none
    skips = [g.sym(k) for k in BACKBONE_FEATURES[encoder][:4]]
    g.skip_layers = [s.layer for s in skips]
    g.encoder_output = x.layer
    g.encoder_layers = list(g.layers)
    if decoder == "U-Net":
        x = _unet(g, x, skips, classes)
    elif decoder == "Linknet":
        x = _linknet(g, x, skips, classes)
    else:
        x = _fpn(g, x, skips, classes)
    g.logits_layer = x.layer
    if activation is None:
        activation = "softmax" if classes > 1 else "sigmoid"
    if temperature is not None:
        x = g.tss(x, temperature)
    else:
        x = g.act(x, activation, activation)
    g.activation = activation
    g.temperature = temperature
    g.encoder, g.decoder, g.classes = encoder, decoder, classes
    g.input_shape = (h, w, c)
    return g
