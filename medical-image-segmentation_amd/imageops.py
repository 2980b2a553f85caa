"""The steps either side of the network on the device, under the names the reference gives them
(/root/reference utils/data.py `normalize_image` :64-80, `one_hot_encode` :31-61, the resizes of
`load_image` :133-138, `augment_dataset` :240-295; utils/predict.py:164-166 collapse + resize
back).  Every function takes and returns torch device tensors (NHWC); numpy inputs are uploaded.
"""
import numpy as np
import torch

from . import lib as L


def _dev():
    if not torch.cuda.is_available():
        raise L.MisegError("image operations run on the GPU; no GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _to_dev(a, dtypes):
    if not isinstance(a, torch.Tensor):
        a = torch.from_numpy(np.ascontiguousarray(a))
    if a.dtype not in dtypes:
        a = a.to(dtypes[-1])
    return a.to(_dev()).contiguous()


def _batched(a, rank):
    return a if a.dim() == rank else a.unsqueeze(0)


def normalize_image(image):
    """Per-image (x - min) / (max - min) over all channels; uint8 or float32 (N,H,W,C) or (H,W,C)
    -> float32 of the same shape."""
    lib = L.load()
    x = _to_dev(image, (torch.uint8, torch.float32))
    xb = _batched(x, 4)
    n = xb.shape[0]
    per = xb[0].numel()
    mm = torch.empty(2 * n, dtype=torch.float32, device=x.device)
    out = torch.empty(xb.shape, dtype=torch.float32, device=x.device)
    dt = 0 if x.dtype == torch.uint8 else 1
    L.check(lib.miseg_image_minmax(xb.data_ptr(), dt, n, per, mm.data_ptr(), _stream()))
    L.check(lib.miseg_image_normalize(xb.data_ptr(), dt, n, per, mm.data_ptr(), out.data_ptr(), _stream()))
    return out.view(x.shape)


def one_hot_encode(mask, classes):
    """Class ids (N,H,W[,1]) or (H,W[,1]) uint8 -> float32 one-hot with `classes` channels."""
    lib = L.load()
    m = _to_dev(mask, (torch.uint8,))
    if m.dim() >= 3 and m.shape[-1] == 1:
        m = m.squeeze(-1)
    out = torch.empty(tuple(m.shape) + (classes,), dtype=torch.float32, device=m.device)
    L.check(lib.miseg_one_hot(m.data_ptr(), out.data_ptr(), m.numel(), classes, _stream()))
    return out


def resize_bilinear(image, shape):
    """tf.image.resize(image, shape, method="bilinear"): uint8/float32 -> float32."""
    lib = L.load()
    x = _to_dev(image, (torch.uint8, torch.float32))
    xb = _batched(x, 4)
    n, h, w, c = xb.shape
    out = torch.empty((n, shape[0], shape[1], c), dtype=torch.float32, device=x.device)
    L.check(lib.miseg_resize_bilinear(xb.data_ptr(), 0 if x.dtype == torch.uint8 else 1, out.data_ptr(), n, h,
                                      w, c, shape[0], shape[1], _stream()))
    return out if x.dim() == 4 else out[0]


def resize_nearest(image, shape, mode="tf"):
    """uint8 nearest-neighbour resize; mode "tf" (tf.image.resize) or "cv2" (cv2.INTER_NEAREST)."""
    lib = L.load()
    x = _to_dev(image, (torch.uint8,))
    squeeze_c = x.dim() == 2
    if squeeze_c:
        x = x.unsqueeze(-1)
    xb = _batched(x, 4)
    n, h, w, c = xb.shape
    out = torch.empty((n, shape[0], shape[1], c), dtype=torch.uint8, device=x.device)
    L.check(lib.miseg_resize_nearest_u8(xb.data_ptr(), out.data_ptr(), n, h, w, c, shape[0], shape[1],
                                        {"tf": 0, "cv2": 1}[mode], _stream()))
    out = out if x.dim() == 4 else out[0]
    return out.squeeze(-1) if squeeze_c else out


def augment(images, masks, params):
    """augment_dataset with the random decisions in `params` (N,8) float32: flip_left_right,
    flip_up_down, brightness delta, contrast factor, hue delta, saturation factor, contrast on,
    hue/saturation on.  images float32 (N,H,W,3); masks float32 (N,H,W,C) or None."""
    lib = L.load()
    x = _to_dev(images, (torch.float32,))
    m = None if masks is None else _to_dev(masks, (torch.float32,))
    p = _to_dev(np.asarray(params, np.float32) if not isinstance(params, torch.Tensor) else params, (torch.float32,))
    n, h, w, _ = x.shape
    xo = torch.empty_like(x)
    mo = None if m is None else torch.empty_like(m)
    ws = torch.empty(3 * n, dtype=torch.float64, device=x.device)
    L.check(lib.miseg_augment(x.data_ptr(), m.data_ptr() if m is not None else None, xo.data_ptr(),
                              mo.data_ptr() if mo is not None else None, n, h, w,
                              m.shape[-1] if m is not None else 0, p.data_ptr(), ws.data_ptr(), _stream()))
    return xo, mo


def random_augment_params(n, rng, threshold=0.5, brightness_delta=0.2, contrast=(0.6, 1.6), hue_delta=0.2,
                          saturation=(0.6, 1.6)):
    """The decisions augment_dataset draws (utils/data.py:267-293), from a numpy Generator: each
    transformation is applied when a uniform draw exceeds `threshold`."""
    p = np.zeros((n, 8), np.float32)
    for i in range(n):
        p[i, 0] = rng.random() > threshold
        p[i, 1] = rng.random() > threshold
        p[i, 2] = rng.uniform(-brightness_delta, brightness_delta) if rng.random() > threshold else 0.0
        p[i, 6] = rng.random() > threshold
        p[i, 3] = rng.uniform(*contrast) if p[i, 6] else 1.0
        hue_on = rng.random() > threshold
        sat_on = rng.random() > threshold
        p[i, 4] = rng.uniform(-hue_delta, hue_delta) if hue_on else 0.0
        p[i, 5] = rng.uniform(*saturation) if sat_on else 1.0
        p[i, 7] = hue_on or sat_on
    return p


def masks_to_original_size(mask_u8, original_shape):
    """cv2.resize(prediction, original_shape[:2][::-1], interpolation=cv2.INTER_NEAREST)
    (utils/predict.py:166) of collapsed uint8 masks (N,H,W,C)."""
    return resize_nearest(mask_u8, original_shape[:2], mode="cv2")
