"""Weight containers for a netspec.Graph: Keras initialisers, get/set in
`get_weights()` order (the order /root/reference utils/model.py:135-141 relies
on when it rebuilds a model at a new input shape)."""
from collections import OrderedDict

import numpy as np


def _uniform(rng, shape, limit):
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def init_weights(graph, seed=None, randomize_bn=False):
    """he_uniform = U(+-sqrt(6/fan_in)), glorot_uniform = U(+-sqrt(6/(fan_in+fan_out)))
    with fans taken from the stored kernel shape as Keras does; biases and BN
    beta/mean zero, gamma/variance one.  `randomize_bn` draws non-trivial BN
    parameters and statistics (used by benchmarks so inference-mode BN does work)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for layer in graph.layers.values():
        if not layer.weights:
            continue
        d = OrderedDict()
        for wname, shape, init in layer.weights:
            if init == "zeros":
                d[wname] = np.zeros(shape, np.float32)
            elif init == "ones":
                d[wname] = np.ones(shape, np.float32)
            else:
                receptive = shape[0] * shape[1]
                fan_in, fan_out = shape[2] * receptive, shape[3] * receptive
                if init == "he_uniform":
                    d[wname] = _uniform(rng, shape, np.sqrt(6.0 / fan_in))
                elif init == "glorot_uniform":
                    d[wname] = _uniform(rng, shape, np.sqrt(6.0 / (fan_in + fan_out)))
                else:
                    raise ValueError("unknown initializer %r" % (init,))
        if randomize_bn:
            if layer.cls == "BatchNormalization":
                c = d["beta"].shape[0]
                if "gamma" in d:
                    d["gamma"] = rng.uniform(0.8, 1.2, c).astype(np.float32)
                d["beta"] = rng.normal(0, 0.1, c).astype(np.float32)
                d["moving_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
                d["moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            elif "bias" in d:
                d["bias"] = rng.normal(0, 0.1, d["bias"].shape).astype(np.float32)
        out[layer.name] = d
    return out


def to_list(graph, weights):
    """Flat list in get_weights() order."""
    return [weights[l][w] for l, w, _, _ in graph.weight_spec()]


def from_list(graph, arrays):
    spec = graph.weight_spec()
    if len(arrays) != len(spec):
        raise ValueError("expected %d weight arrays, got %d" % (len(spec), len(arrays)))
    out = OrderedDict()
    for (l, w, shape, _), a in zip(spec, arrays):
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError("layer %s weight %s: expected shape %r, got %r" % (l, w, shape, a.shape))
        out.setdefault(l, OrderedDict())[w] = a
    return out
