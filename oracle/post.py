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
