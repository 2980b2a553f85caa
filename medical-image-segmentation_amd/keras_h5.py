"""Keras HDF5 model files (`model.save("x.h5")` of tf.keras 2.x, loaded by the reference at
/root/reference utils/model.py:265 and written by its ModelCheckpoint, model_train.py:262) to and
from this package's Model, without h5py or TensorFlow.

File layout (Keras `save_model_to_hdf5` / `save_weights_to_hdf5_group`):
  /            attrs  keras_version, backend, model_config (JSON), training_config (JSON)
  /model_weights      attrs  layer_names [, layer_names0, layer_names1, ... when split], backend,
                             keras_version
  /model_weights/<layer name>            attrs  weight_names (e.g. b"conv0/kernel:0")
  /model_weights/<layer name>/<weight name>     one dataset per weight, HWIO kernels
A weights-only file (`model.save_weights`) has the content of /model_weights at the root.
The architecture is recognised from the layer names in model_config (segmentation_models /
classification_models / keras_applications naming), then rebuilt here and checked weight by weight.
"""
import json
from collections import OrderedDict

import numpy as np

from . import h5io, netspec

_W_ORDER = ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_variance")


def _text(v):
    if isinstance(v, (bytes, np.bytes_)):
        return bytes(v).decode("utf-8")
    return v


def infer_architecture(config):
    """(encoder, decoder, input_shape, classes, activation, temperature) from a Keras functional
    model config (the dict under 'config' or the whole model_config)."""
    cfg = config.get("config", config)
    layers = cfg["layers"]
    names = [l.get("name") or l["config"]["name"] for l in layers]
    nameset = set(names)
    by_name = {n: l for n, l in zip(names, layers)}
    shape = layers[0]["config"]["batch_input_shape"][1:]

    def count(fmt, start=1):
        n = 0
        while fmt % (start + n) in nameset:
            n += 1
        return n
    if "stage1_unit1_bn1" in nameset:
        units = tuple(count("stage%d_unit%%d_bn1" % s) for s in (1, 2, 3, 4))
        bott = "stage1_unit1_conv3" in nameset      # resnet34 and resnet50 share their unit counts
        enc = {v: k for k, v in netspec.RESNET_UNITS.items()
               if (k in netspec.RESNET_BOTTLENECK) == bott}.get(units)
    elif "conv2_block1_0_bn" in nameset:
        blocks = tuple(count("conv%d_block%%d_0_bn" % s) for s in (2, 3, 4, 5))
        enc = {v: k for k, v in netspec.DENSENET_LAYERS.items()}.get(blocks)
    elif "block1_conv1" in nameset and "block5_pool" in nameset:
        blocks = tuple(count("block%d_conv%%d" % s) for s in (1, 2, 3, 4, 5))
        enc = {v: k for k, v in netspec.VGG_BLOCKS.items()}.get(blocks)
    else:
        enc = None
    if enc is None:
        raise ValueError("the encoder of this model is not one of %s"
                         % sorted(list(netspec.RESNET_UNITS) + list(netspec.DENSENET_LAYERS) +
                                  list(netspec.VGG_BLOCKS)))
    if "decoder_stage0a_transpose" in nameset:
        dec, head = "U-Net", "final_conv"
    elif "decoder_stage0b_transpose" in nameset:
        dec, head = "Linknet", "final_conv"
    elif "fpn_stage_p5_add" in nameset or "fpn_stage_p5_conv" in nameset:
        dec, head = "FPN", "head_conv"
    else:
        raise ValueError("the decoder of this model is not U-Net (transpose), Linknet or FPN")
    classes = int(by_name[head]["config"]["filters"])
    last = layers[-1]
    temperature = None
    if last["class_name"] == "TemperatureScaledSoftmax":
        activation, temperature = "softmax", float(last["config"].get("temperature", 1))
    elif last["class_name"] == "Activation":
        activation = last["config"]["activation"]
    else:
        activation = "softmax" if classes > 1 else "sigmoid"
    return {"encoder": enc, "decoder": dec, "input_shape": tuple(int(s) for s in shape),
            "classes": classes, "activation": activation, "temperature": temperature,
            "name": cfg.get("name")}


def _read_layer_weights(group, layer):
    """{weight: array} of one layer group; weights are matched by the last path component of the
    Keras weight name ('stage1_unit1_bn1/gamma:0' -> gamma)."""
    names = group.attrs.get("weight_names")
    out = OrderedDict()
    if names is None:
        found = dict(group.visit_datasets())
        names = list(found)
    for raw in np.asarray(names).ravel():
        wname = _text(raw)
        key = wname.split("/")[-1].split(":")[0]
        out[key] = np.asarray(group[wname].read(), dtype=np.float32)
    return out


def load_keras_h5(path, model_cls=None):
    from .model import Model
    f = h5io.File(str(path))
    cfg = f.attrs.get("model_config")
    if cfg is None:
        raise ValueError("%s holds weights only (no model_config attribute): build the model with "
                         "make_model and use load_weights_h5" % path)
    arch = infer_architecture(json.loads(_text(cfg)))
    g = netspec.build_graph(arch["encoder"], arch["decoder"], arch["input_shape"], arch["classes"],
                            activation=arch["activation"], temperature=arch["temperature"], name=arch["name"])
    model = (model_cls or Model)(g, weights=read_weights(f, g))
    # a model saved with a frozen encoder comes back frozen: every weighted backbone layer non-trainable
    by_name = {(l.get("name") or l["config"]["name"]): l["config"] for l in
               json.loads(_text(cfg)).get("config", {}).get("layers", [])}
    weighted = [n for n in g.encoder_layers if g.layers[n].weights]
    model.encoder_freeze = bool(weighted) and all(by_name.get(n, {}).get("trainable", True) is False for n in weighted)
    return model


def read_weights(f, graph):
    root = f["model_weights"] if "model_weights" in f else f
    weights = OrderedDict()
    for l in graph.layers.values():
        if not l.weights:
            continue
        try:
            grp = root[l.name]
        except KeyError:
            raise ValueError("layer %r is missing from the file" % l.name) from None
        got = _read_layer_weights(grp, l)
        d = OrderedDict()
        for wn, shape, _ in l.weights:
            if wn not in got:
                raise ValueError("weight %s/%s is missing from the file" % (l.name, wn))
            if tuple(got[wn].shape) != tuple(shape):
                raise ValueError("%s/%s: file has shape %r, the graph needs %r"
                                 % (l.name, wn, tuple(got[wn].shape), tuple(shape)))
            d[wn] = got[wn]
        weights[l.name] = d
    return weights


def load_weights_h5(model, path):
    f = h5io.File(str(path))
    model.weights = read_weights(f, model.graph)
    model._invalidate()


def _keras_layer_config(l):
    """Keras `get_config()` of the layer, complete enough for `from_config`."""
    c = dict(l.config, name=l.name)
    if l.cls in ("Conv2D", "Conv2DTranspose"):
        c["kernel_size"] = [c["kernel_size"]] * 2
        c["strides"] = [c["strides"]] * 2
        c["kernel_initializer"] = {"class_name": "HeUniform" if c.get("kernel_initializer") == "he_uniform"
                                   else "GlorotUniform", "config": {"seed": None}}
        c.update({"dilation_rate": [1, 1], "groups": 1, "activation": c.get("activation", "linear"),
                  "data_format": "channels_last", "trainable": True, "dtype": "float32"})
    elif l.cls == "ZeroPadding2D":
        p = c["padding"]
        c["padding"] = [[p, p], [p, p]]
    elif l.cls == "MaxPooling2D" or l.cls == "AveragePooling2D":
        c["pool_size"] = [c["pool_size"]] * 2
        c["strides"] = [c["strides"]] * 2
        c.setdefault("padding", "valid")
    elif l.cls == "UpSampling2D":
        c["size"] = [c["size"]] * 2
    elif l.cls == "BatchNormalization":
        c["axis"] = [3]
    elif l.cls == "InputLayer":
        c.update({"dtype": "float32", "sparse": False, "ragged": False})
    return c


def keras_model_config(model):
    g = model.graph
    layers = []
    # encoder_freeze (utils/model.py:179,191,203 -> segmentation_models freeze_model): the backbone's layers
    # carry trainable = False in the saved config
    frozen = set(g.encoder_layers) if getattr(model, "encoder_freeze", False) else set()
    for l in g.layers.values():
        cfg = _keras_layer_config(l)
        if l.name in frozen and l.cls != "InputLayer":
            cfg["trainable"] = False
        layers.append({"class_name": l.cls, "config": cfg, "name": l.name,
                       "inbound_nodes": [[[n, 0, 0, {}] for n in l.inbound]] if l.inbound else []})
    first, last = layers[0]["name"], layers[-1]["name"]
    return {"class_name": "Functional",
            "config": {"name": model.name, "layers": layers, "input_layers": [[first, 0, 0]],
                       "output_layers": [[last, 0, 0]]}}


def save_keras_h5(model, path, keras_version="2.10.0", backend="tensorflow"):
    """Write `model` in the Keras HDF5 layout.  Attribute kinds follow what h5py produces for the
    values Keras passes (Python bytes -> variable-length strings; lists of bytes -> fixed-length
    string arrays, split when larger than 64000 bytes)."""
    w = h5io.Writer()
    kv, be = keras_version.encode(), backend.encode()
    w.attr("/", "keras_version", kv)
    w.attr("/", "backend", be)
    w.attr("/", "model_config", json.dumps(keras_model_config(model)).encode("utf-8"))
    names = [l.name for l in model.graph.layers.values()]
    _split_attr(w, "/model_weights", "layer_names", [n.encode() for n in names])
    w.attr("/model_weights", "backend", be)
    w.attr("/model_weights", "keras_version", kv)
    for l in model.graph.layers.values():
        gpath = "/model_weights/" + l.name
        w.group(gpath)
        wnames = []
        for wn, _, _ in l.weights:
            full = "%s/%s:0" % (l.name, wn)
            wnames.append(full.encode())
            w.dataset(gpath + "/" + full, np.asarray(model.weights[l.name][wn], np.float32))
        _split_attr(w, gpath, "weight_names", wnames)
    w.save(path)


def _split_attr(w, gpath, name, values, limit=64000 - 1024):
    if not values:
        w.attr(gpath, name, np.zeros((0,), dtype="S1"))
        return
    arr = np.asarray(values)
    if arr.nbytes <= limit:
        w.attr(gpath, name, arr)
        return
    parts = 2
    while max(c.nbytes for c in np.array_split(arr, parts)) > limit:
        parts += 1
    for i, chunk in enumerate(np.array_split(arr, parts)):
        w.attr(gpath, "%s%d" % (name, i), chunk)
