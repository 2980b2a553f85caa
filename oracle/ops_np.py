"""numpy fp32 restatement of the layer arithmetic the reference reaches through
tf.keras (reference call sites: utils/model.py:176-211 builders,
utils/predict.py:159 forward).  TEST INFRASTRUCTURE: see oracle/__init__.py.

Every function takes and returns NHWC float32 numpy arrays.  Semantics follow
TensorFlow 2.10 / Keras 2.10 layers as pinned by environment.yml:42,90.
"""
import numpy as np

name = "numpy"


def asarray(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def to_numpy(x):
    return np.asarray(x)


def zero_pad(x, p):
    """keras.layers.ZeroPadding2D(p): symmetric spatial zero padding."""
    if p == 0:
        return x
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))


def conv2d(x, w, stride=1, pad=0, bias=None):
    """Cross-correlation, kernel HWIO, explicit symmetric zero padding `pad`,
    then 'valid' with `stride` (Keras Conv2D after ZeroPadding2D; with
    stride 1 and pad=(k-1)/2 this is Keras padding='same')."""
    kh, kw, cin, cout = w.shape
    xp = zero_pad(x, pad)
    n, hp, wp, _ = xp.shape
    ho = (hp - kh) // stride + 1
    wo = (wp - kw) // stride + 1
    out = np.zeros((n * ho * wo, cout), dtype=np.float32)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, dy:dy + (ho - 1) * stride + 1:stride,
                       dx:dx + (wo - 1) * stride + 1:stride, :]
            out += patch.reshape(-1, cin) @ w[dy, dx]
    out = out.reshape(n, ho, wo, cout)
    if bias is not None:
        out = out + bias.astype(np.float32)
    return out


def conv2d_transpose_4x4s2(x, w):
    """Keras Conv2DTranspose(filters, 4, strides=2, padding='same',
    use_bias=False); kernel stored (kh, kw, Cout, Cin).  Output is 2H x 2W.

    It is the transpose of a 4x4 stride-2 'same' convolution, whose padding is
    one pixel before and one after: input pixel i scatters kernel tap k to
    output 2i - 1 + k."""
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert (kh, kw) == (4, 4) and cin2 == cin
    full = np.zeros((n, 2 * h + 2, 2 * wd + 2, cout), dtype=np.float32)
    flat = x.reshape(-1, cin)
    for a in range(4):
        for b in range(4):
            contrib = (flat @ w[a, b].T).reshape(n, h, wd, cout)
            full[:, a:a + 2 * h:2, b:b + 2 * wd:2, :] += contrib
    return full[:, 1:2 * h + 1, 1:2 * wd + 1, :]


def batchnorm(x, gamma, beta, mean, var, eps):
    """Inference-mode BatchNormalization(axis=3) with moving statistics.
    gamma is None for scale=False layers (ResNet `bn_data`)."""
    inv = 1.0 / np.sqrt(var.astype(np.float32) + np.float32(eps))
    if gamma is not None:
        inv = inv * gamma
    return (x - mean) * inv + beta


def batchnorm_train(x, gamma, beta, eps):
    """Training-mode BatchNormalization(axis=3): normalise with the batch mean and the
    biased batch variance (float64 accumulation); returns (y, mean, variance)."""
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 1, 2))
    var = ((x64 - mean) ** 2).mean(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    y = (x64 - mean) * inv + beta
    return y.astype(np.float32), mean.astype(np.float32), var.astype(np.float32)


def relu(x):
    return np.maximum(x, 0)


def maxpool3x3s2_pad1(x):
    """ZeroPadding2D(1) then MaxPooling2D(3, strides=2, 'valid')."""
    xp = zero_pad(x, 1)
    n, hp, wp, c = xp.shape
    ho = (hp - 3) // 2 + 1
    wo = (wp - 3) // 2 + 1
    out = None
    for dy in range(3):
        for dx in range(3):
            v = xp[:, dy:dy + 2 * (ho - 1) + 1:2, dx:dx + 2 * (wo - 1) + 1:2, :]
            out = v if out is None else np.maximum(out, v)
    return out


def maxpool2x2(x):
    """MaxPooling2D(2, strides=2), 'valid' (keras_applications VGG block pools)."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    return x[:, :2 * h2, :2 * w2, :].reshape(n, h2, 2, w2, 2, c).max(axis=(2, 4))


def avgpool2x2(x):
    """AveragePooling2D(2, strides=2), 'valid'."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    v = x[:, :2 * h2, :2 * w2, :].reshape(n, h2, 2, w2, 2, c)
    return v.mean(axis=(2, 4), dtype=np.float32)


def upsample_nearest(x, s):
    """UpSampling2D(size=s, interpolation='nearest'): out[y,x] = in[y//s, x//s]."""
    return np.repeat(np.repeat(x, s, axis=1), s, axis=2)


def upsample_bilinear2x(x):
    """UpSampling2D(2, interpolation='bilinear') == tf.image.resize with
    half-pixel centres: src = (dst + 0.5)/2 - 0.5, clamped at the borders."""
    def axis_coeffs(n_in):
        dst = np.arange(2 * n_in, dtype=np.float64)
        src = (dst + 0.5) / 2.0 - 0.5
        src = np.clip(src, 0.0, n_in - 1)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        f = (src - i0).astype(np.float32)
        return i0, i1, f
    n, h, w, c = x.shape
    y0, y1, fy = axis_coeffs(h)
    x0, x1, fx = axis_coeffs(w)
    rows = x[:, y0] * (1 - fy)[None, :, None, None] + x[:, y1] * fy[None, :, None, None]
    out = rows[:, :, x0] * (1 - fx)[None, None, :, None] + rows[:, :, x1] * fx[None, None, :, None]
    return out.astype(np.float32)


def concat(xs):
    return np.concatenate(xs, axis=-1)


def add(a, b):
    return a + b


def softmax(x, temperature=1.0):
    """tf.nn.softmax over the channel axis; logits are divided by the
    temperature first (utils/model.py:21-23)."""
    z = x / np.float32(temperature)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))
