"""numpy restatement of the in-repo steps either side of the network forward.
TEST INFRASTRUCTURE: see oracle/__init__.py.

* normalize_image      utils/data.py:64-80
* one_hot_encode       utils/data.py:50-54 (tensor branch)
* collapse_probabilities  utils/utils.py:50-68
* patch ranges         utils/predict.py:44-61
"""
import numpy as np


def normalize_image(image):
    """(x - min) / (max - min) over the whole image, float32."""
    image = np.asarray(image).astype(np.float32)
    lo, hi = image.min(), image.max()
    return (image - lo) / (hi - lo)


def one_hot_encode(mask, classes):
    """(H,W,1) or (H,W) class ids -> (H,W,classes) float32 one-hot."""
    ids = np.asarray(mask).astype(np.int32).reshape(mask.shape[0], mask.shape[1])
    out = np.zeros(ids.shape + (classes,), np.float32)
    for c in range(classes):
        out[..., c] = ids == c
    return out


def collapse_probabilities(prediction, pixel_intensity=127):
    """Channel i becomes `pixel_intensity` where it is STRICTLY greater than
    every other channel, else 0; ties give an all-zero pixel.  The reference
    updates channels in place, one after another, so channel i is compared
    with the already-collapsed values of channels j < i; that is reproduced
    here because it changes results when probabilities exceed the intensity
    or tie at zero."""
    pred = np.array(prediction, dtype=np.float32, copy=True)
    classes = pred.shape[-1]
    for i in range(classes):
        win = np.ones(pred.shape[:-1], dtype=bool)
        for j in range(classes):
            if j != i:
                win &= pred[..., i] > pred[..., j]
        pred[..., i] = np.where(win, pixel_intensity, 0)
    return pred.astype(np.uint8)


def collapse_simple(prediction, pixel_intensity=127):
    """Strict-argmax form: what the in-place loop computes whenever all
    inputs are probabilities in [0, 1] and pixel_intensity > 1."""
    pred = np.asarray(prediction, np.float32)
    out = np.zeros(pred.shape, np.uint8)
    classes = pred.shape[-1]
    for i in range(classes):
        win = np.ones(pred.shape[:-1], dtype=bool)
        for j in range(classes):
            if j != i:
                win &= pred[..., i] > pred[..., j]
        out[..., i] = np.where(win, pixel_intensity, 0)
    return out


def patch_ranges(size, patch):
    """Tile ranges along one axis: non-overlapping, the last one shifted back
    to end at the border."""
    out, start = [], 0
    for _ in range(int(np.ceil(size / patch))):
        end = start + patch
        if end > size:
            end = size
            start = end - patch
        out.append([start, end])
        start = end
    return out


# ---- resizing and augmentation: TensorFlow / OpenCV semantics restated from their documentation.
# PARITY UNPINNED: neither library is installed here and the reference holds no image fixtures.
def resize_bilinear_tf(image, out_h, out_w):
    """tf.image.resize(image, (out_h, out_w), method="bilinear") (utils/data.py:138): half-pixel
    centres, no antialiasing, float32 result.  image: (H,W,C)."""
    img = np.asarray(image).astype(np.float32)
    h, w = img.shape[:2]

    def axis(n_in, n_out):
        f = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        fl = np.floor(f)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(f).astype(np.int64), n_in - 1)
        return lo, hi, (f - fl).astype(np.float32)
    y0, y1, ly = axis(h, out_h)
    x0, x1, lx = axis(w, out_w)
    lx = lx[None, :, None]
    ly = ly[:, None, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * lx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * lx
    return top + (bot - top) * ly


def resize_nearest(image, out_h, out_w, mode="tf"):
    """mode "tf": tf.image.resize(method="nearest") (utils/data.py:136): floor((dst+0.5)*scale);
    mode "cv2": cv2.resize(..., interpolation=cv2.INTER_NEAREST) (utils/predict.py:166):
    floor(dst*scale).  image: (H,W[,C])."""
    img = np.asarray(image)
    h, w = img.shape[:2]
    off = np.float32(0.5 if mode == "tf" else 0.0)
    ys = np.minimum(np.floor((np.arange(out_h, dtype=np.float32) + off) * np.float32(h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor((np.arange(out_w, dtype=np.float32) + off) * np.float32(w / out_w)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def rgb_to_hsv(rgb):
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    v = np.max(rgb, axis=-1)
    rng = v - np.min(rgb, axis=-1)
    s = np.where(v > 0, rng / np.where(v > 0, v, 1), 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        norm = 1.0 / (6.0 * rng)
        h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
    h = np.where(rng <= 0, 0.0, h)
    h = np.where(h < 0, h + 1.0, h)
    return np.stack([h, s, v], axis=-1).astype(np.float32)


def hsv_to_rgb(hsv):
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    c = s * v
    m = v - c
    dh = h * 6.0
    fm = np.mod(dh, 2.0)
    x = c * (1.0 - np.abs(fm - 1.0))
    cat = dh.astype(np.int64)
    z = np.zeros_like(c)
    rr = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4], [c, x, z, z, x], c)
    gg = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4], [x, c, c, x, z], z)
    bb = np.select([cat == 0, cat == 1, cat == 2, cat == 3, cat == 4], [z, z, x, c, c], x)
    return np.stack([rr + m, gg + m, bb + m], axis=-1).astype(np.float32)


def augment(image, mask, flip_lr=False, flip_ud=False, brightness=0.0, contrast=None, hue=None,
            saturation=None):
    """augment_dataset (utils/data.py:240-295) for one float32 RGB image with the random draws
    given: flips, adjust_brightness (x + delta), adjust_contrast ((x - mean_c) * f + mean_c),
    adjust_hue, adjust_saturation (through HSV)."""
    img = np.asarray(image, np.float32)
    msk = None if mask is None else np.asarray(mask)
    if flip_lr:
        img = img[:, ::-1]
        msk = None if msk is None else msk[:, ::-1]
    if flip_ud:
        img = img[::-1]
        msk = None if msk is None else msk[::-1]
    img = img + np.float32(brightness)
    if contrast is not None:
        mean = img.astype(np.float64).mean(axis=(0, 1)).astype(np.float32)
        img = (img - mean) * np.float32(contrast) + mean
    if hue is not None or saturation is not None:
        hsv = rgb_to_hsv(img)
        if hue is not None:
            hsv[..., 0] = np.mod(hsv[..., 0] + np.float32(hue), 1.0)
        if saturation is not None:
            hsv[..., 1] = np.clip(hsv[..., 1] * np.float32(saturation), 0.0, 1.0)
        img = hsv_to_rgb(hsv)
    return np.ascontiguousarray(img, np.float32), (None if msk is None else np.ascontiguousarray(msk))
