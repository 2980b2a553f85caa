"""Drop-in for the prediction loop of the reference's `utils/predict.py`
(/root/reference utils/predict.py:17-72,75-166,277-286): load, resize, normalise, forward (whole
image or the reference's border-anchored patches), collapse to class masks, resize back to the
original size, write PNGs.  Decode and PNG encode run on the host (Pillow); everything between
them runs on the GPU.  The OpenCV contour analysis that the reference can chain after the mask
(`analyze_contours`, `reclassify`, `remove_artifacts`, `use_bias*`) is outside this package.
"""
import os
import shutil
import time
from pathlib import Path

import numpy as np
import torch

from utils.data import list_files, load_image, reset_class_values
from utils.model import get_model_input_shape, load_model
from utils.utils import color_classes
from miseg_amd import imageops as _ops, shard as _shard
from miseg_amd.model import Model as _Model


def halo_ranges(size, patch, halo):
    """Windows of `patch` pixels along one axis of `size` pixels whose inner parts tile the axis:
    a list of (window start, paste from, paste to).  A window gives up `halo` pixels at every edge
    that is not the image border, so each output pixel comes from a window that saw at least `halo`
    pixels of context on every side (or the image border).  The last window is shifted back to end
    at the border, as the reference's tiles are (/root/reference utils/predict.py:47-61)."""
    if halo < 0 or 2 * halo >= patch:
        raise ValueError("halo must satisfy 0 <= 2 * halo < patch (%d), got %d" % (patch, halo))
    if size < patch:
        raise ValueError("the image (%d) is smaller than the patch (%d)" % (size, patch))
    out, lo = [], 0
    while lo < size:
        start = min(max(lo - halo, 0), size - patch)
        hi = size if start + patch >= size else start + patch - halo
        out.append((start, lo, hi))
        lo = hi
    return out


def patch_predict(model, image, patch_shape, halo=0):
    """Forward `image` (H,W,3 float32, host or device) in patches of the model's input size.
    Returns the (H,W,classes) float32 probabilities as a device tensor.

    halo = 0 (default) is the reference's `patch_predict`: non-overlapping patches, the last patch
    of each axis shifted back to end at the border, later patches overwriting earlier ones where
    they overlap; every patch is forwarded without context, so predictions change at patch seams.
    halo > 0 (an option the reference does not have) forwards overlapping windows and keeps only
    the part of each that lies at least `halo` pixels inside it (see `halo_ranges`).

    One deliberate difference: the reference pastes into a buffer hard-coded to 8 channels
    (/root/reference utils/predict.py:63), which only works for a model with exactly 8 classes (numpy
    cannot broadcast a (h, w, 3) patch into an (h, w, 8) slice, so its 3-class models raise there).
    Here the buffer has the model's own number of classes, which is the same result wherever the
    reference's line runs at all."""
    input_shape = get_model_input_shape(model)
    if tuple(input_shape) != tuple(patch_shape):
        raise ValueError("The model's input shape does not match the patch shape: model '%s' != patch '%s'."
                         % (input_shape, patch_shape))
    if not isinstance(image, torch.Tensor):
        image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
    image = image.cuda()
    h, w = image.shape[:2]
    out = None
    if halo:
        for ys, y0, y1 in halo_ranges(h, patch_shape[0], halo):
            for xs, x0, x1 in halo_ranges(w, patch_shape[1], halo):
                window = image[ys:ys + patch_shape[0], xs:xs + patch_shape[1]]
                pr = model(window.unsqueeze(0).contiguous(), training=False).tensor[0]
                if out is None:
                    out = torch.zeros((h, w, pr.shape[-1]), dtype=torch.float32, device=pr.device)
                out[y0:y1, x0:x1] = pr[y0 - ys:y1 - ys, x0 - xs:x1 - xs]
        return out
    for y0, y1 in _shard.tile_ranges(h, patch_shape[0]):
        for x0, x1 in _shard.tile_ranges(w, patch_shape[1]):
            pr = model(image[y0:y1, x0:x1].unsqueeze(0).contiguous(), training=False).tensor[0]
            if out is None:
                out = torch.zeros((h, w, pr.shape[-1]), dtype=torch.float32, device=pr.device)
            out[y0:y1, x0:x1] = pr
    return out


def files_of_rank(images, shard=None):
    """The image files this process segments: `images` is one file or a directory; of the sorted,
    de-duplicated list (/root/reference utils/predict.py:125-131,152) rank r of w takes files r, r + w,
    r + 2 w, ...  Returns (files, rank, world); `shard=None` reads RANK / WORLD_SIZE."""
    if not isinstance(images, str):
        raise ValueError("`images` must be a `str`. Given `%s`." % type(images))
    if Path(images).is_dir():
        files = list_files(images, as_numpy=True)
    elif Path(images).is_file():
        files = [images]
    else:
        raise FileNotFoundError("The directory or file was not found at `%s`." % images)
    if shard is None:
        shard = (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
    rank, world = shard
    files = sorted(set(str(f) for f in files))
    return [files[i] for i in _shard.shard_indices(len(files), rank, world)], rank, world


def predict(model, images, normalize=True, input_shape=None, copy_images=False, grayscale=False,
            analyze_contours=False, output_predictions="predictions", output_contour_analysis=None, bboxes=None,
            classify_agnor=False, decision_tree_model_path="agnor_classifier.joblib", record_id=None,
            record_class=None, measures_only=False, current_time=time.strftime("%Y%m%d%H%M%S"),
            use_bias_layer=False, use_bias=False, reclassify=False, remove_artifacts=False, shard=None):
    """Segment one image file or every image of a directory and write `<stem>_prediction.png`
    (class colours, or class indices with `grayscale`) under `output_predictions`.

    The reference walks the sorted files one by one on one GPU (/root/reference utils/predict.py:152).
    Images are independent given the weights, so with one process per GPU each process takes every
    world-th file of the same sorted list and nothing is exchanged: `shard=(rank, world)`, by default
    what `torch.distributed.run` left in RANK / WORLD_SIZE (predict.py --gpu 0,1,... starts it)."""
    if analyze_contours or use_bias or use_bias_layer or reclassify or remove_artifacts or classify_agnor:
        raise NotImplementedError("contour analysis and Papanicolaou post-processing are OpenCV tooling on the "
                                  "CPU and are not part of this package; run them on the written masks")
    files, rank, world = files_of_rank(images, shard)
    if isinstance(model, str):
        if world > 1 and torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
        model = load_model(model_path=model, input_shape=input_shape)
    elif not isinstance(model, _Model):
        raise ValueError("`model` must be a `str` or a model object. Given `%s`." % type(model))
    if not input_shape:
        input_shape = get_model_input_shape(model)
    out_dir = Path(output_predictions)
    out_dir.mkdir(exist_ok=True, parents=True)
    from PIL import Image
    for file in files:
        image, original_shape = load_image(image_path=file, normalize=normalize, shape=input_shape[:2],
                                           return_original_shape=True)
        image = image.float()
        if tuple(image.shape) != tuple(input_shape):
            probs = patch_predict(model, image, input_shape)
            mask = _ops_collapse(probs)
        else:
            mask = torch.from_numpy(model.predict_masks(image.unsqueeze(0), pixel_intensity=127)[0]).cuda()
        mask = _ops.masks_to_original_size(mask.unsqueeze(0), original_shape)[0].cpu().numpy()
        if measures_only:
            continue
        picture = reset_class_values(mask) if grayscale else color_classes(mask.copy())
        Image.fromarray(picture).save(str(out_dir.joinpath("%s_prediction.png" % Path(file).stem)))
        if copy_images:
            shutil.copyfile(str(file), str(out_dir.joinpath(Path(file).name)))


def _ops_collapse(probs):
    """Strict-argmax collapse of (H,W,C) device probabilities to a 0/127 uint8 mask."""
    from miseg_amd import lib as L
    lib = L.load()
    out = torch.empty(probs.shape, dtype=torch.uint8, device=probs.device)
    L.check(lib.miseg_collapse_u8(probs.contiguous().data_ptr(), out.data_ptr(), probs.shape[0] * probs.shape[1],
                                  probs.shape[2], 127, torch.cuda.current_stream().cuda_stream))
    return out
