"""Drop-in for the reference's `utils/data.py` (image loading, the input contract and the
training/validation datasets, /root/reference utils/data.py:12-374) without TensorFlow: files are
decoded with Pillow on the host, everything after the decode (resize, augmentation, min-max
normalisation, one-hot masks) runs on the GPU through `miseg_amd.imageops`, and a dataset is a
plain Python iterable of (images, masks) device-tensor batches that `Model.fit` / `Model.evaluate`
consume.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miseg_amd  # noqa: E402,F401
from miseg_amd import imageops as _ops  # noqa: E402

SUPPORTED_IMAGE_TYPES = [".tif", ".tiff", ".png", ".jpg", ".jpeg", ".bmp", ".BMP"]


def reset_class_values(mask):
    """(H,W,C) mask with one channel per class -> (H,W) uint8 class indices (later channels win)."""
    out = np.zeros(mask.shape[:2], dtype=np.uint8)
    for c in range(mask.shape[-1]):
        out[mask[:, :, c] > 0] = c
    return out


def one_hot_encode(image, classes, as_numpy=False):
    """numpy input: uint8 (H,W,classes) with channel k set where the mask holds its k-th distinct
    value; device tensor input: tf.one_hot semantics as float32."""
    if isinstance(image, np.ndarray):
        ids = image[..., 0] if image.ndim == 3 else image
        out = np.zeros(ids.shape[:2] + (classes,), dtype=np.uint8)
        for k, value in enumerate(np.unique(ids)):
            if k < classes:
                out[:, :, k][ids == value] = 1
        return out
    if isinstance(image, torch.Tensor):
        enc = _ops.one_hot_encode(image.to(torch.uint8), classes)
        return enc.cpu().numpy() if as_numpy else enc
    raise TypeError("Argument `image` should be `np.ndarray` or a device tensor. Given `%s`." % type(image))


def normalize_image(image):
    """Min-max normalisation of one image to [0, 1] as float32, on the device."""
    return _ops.normalize_image(image)


def _decode(path, as_gray):
    from PIL import Image
    with Image.open(str(path)) as im:
        im = im.convert("L" if as_gray else "RGB")
        a = np.asarray(im, dtype=np.uint8)
    return a[:, :, None] if as_gray else a


def load_image(image_path, shape=None, normalize=False, as_gray=False, as_numpy=False,
               return_original_shape=False, dataset_branch=False):
    """Decode one image file; optionally resize to `shape` (H, W) — nearest for files whose name
    contains "mask", bilinear otherwise — and min-max normalise.  Returns a device tensor (uint8 if
    untouched, float32 after a bilinear resize or normalisation) or, with as_numpy, an ndarray.
    `dataset_branch` selects the behaviour of the reference's tensor-path branch
    (/root/reference utils/data.py:155-173), the one its tf.data training pipeline goes through:
    every resize is nearest-neighbour, images included."""
    if isinstance(image_path, bytes):
        image_path = image_path.decode("utf-8")
    if not isinstance(image_path, (str, Path)):
        raise TypeError("Object `%r` is not supported." % (image_path,))
    path = Path(image_path)
    if not path.is_file():
        raise FileNotFoundError("The file `%s` was not found." % path)
    if path.suffix not in SUPPORTED_IMAGE_TYPES:
        raise TypeError("Image type `%s` not supported. The supported types are `%s`."
                        % (path.suffix, ", ".join(SUPPORTED_IMAGE_TYPES)))
    raw = _decode(path, as_gray)
    original_shape = tuple(raw.shape)
    image = torch.from_numpy(raw).cuda()
    if shape and tuple(shape) != tuple(raw.shape[:2]):
        if dataset_branch or "mask" in path.stem:
            image = _ops.resize_nearest(image, tuple(shape), mode="tf")
        else:
            image = _ops.resize_bilinear(image, tuple(shape))
    if normalize:
        image = _ops.normalize_image(image)
    if as_numpy:
        image = image.cpu().numpy()
        if as_gray:
            image = image[:, :, 0]
    return (image, original_shape) if return_original_shape else image


def list_files(files_path, as_numpy=False, file_types=SUPPORTED_IMAGE_TYPES, seed=1234):
    """Image files directly under `files_path`: sorted (as_numpy) or in a seeded shuffle that is
    the same for an images directory and its masks directory when their file names sort alike."""
    root = Path(files_path)
    if not root.is_dir():
        raise FileNotFoundError("The directory `%s` does not exist." % files_path)
    found = sorted(str(p) for p in root.iterdir() if p.is_file() and p.suffix in file_types)
    if not found:
        raise RuntimeError("No files were found at `%s`." % files_path)
    if as_numpy:
        return found
    order = np.random.default_rng(seed).permutation(len(found))
    return [found[i] for i in order]


def load_dataset_files(image_path, mask_path, shape, classes, mask_one_hot_encoded=True):
    image = load_image(image_path, shape=shape, normalize=False, dataset_branch=True)
    mask = load_image(mask_path, shape=shape, as_gray=True, dataset_branch=True)
    if mask_one_hot_encoded:
        mask = one_hot_encode(mask, classes=classes)
    return image, mask


def augment_dataset(image, mask, threshold=0.5, brightness_delta=0.2, contrast_lower=0.6, contrast_upper=1.6,
                    hue_delta=0.2, saturation_lower=0.6, saturation_upper=1.6, rng=None):
    """Random flips, brightness, contrast, hue and saturation of one image (and the flips of its
    mask); every transformation is applied when a uniform draw exceeds `threshold`."""
    rng = rng or np.random.default_rng()
    params = _ops.random_augment_params(1, rng, threshold, brightness_delta, (contrast_lower, contrast_upper),
                                        hue_delta, (saturation_lower, saturation_upper))
    img, msk = _ops.augment(image.float().unsqueeze(0), mask.float().unsqueeze(0), params)
    return img[0], msk[0]


class Dataset:
    """Iterable of (images, masks) batches with the options of the reference's `load_dataset`:
    files are listed once, decoded on first use and cached in host memory as uint8 (the reference
    caches its file list and decodes through tf.data), augmented per sample, normalised per image,
    batched, optionally shuffled at batch level and repeated."""

    def __init__(self, images, masks, batch_size, shape, classes, one_hot, repeat, shuffle, augment, seed=1234,
                 rank=0, world=1):
        if len(images) != len(masks):
            raise ValueError("%d images but %d masks" % (len(images), len(masks)))
        self.images, self.masks = images[rank::world], masks[rank::world]
        self.batch_size, self.shape, self.classes, self.one_hot = batch_size, tuple(shape), classes, one_hot
        self.repeat, self.shuffle, self.augment = repeat, shuffle, augment
        self.rng = np.random.default_rng(seed + rank)
        self._cache = {}

    def __len__(self):
        if self.repeat:
            raise TypeError("a repeating dataset has no length")
        return -(-len(self.images) // self.batch_size)

    def _sample(self, i):
        if i not in self._cache:
            img, msk = load_dataset_files(self.images[i], self.masks[i], self.shape, self.classes, self.one_hot)
            self._cache[i] = (img.cpu(), msk.cpu())
        img, msk = self._cache[i]
        return img.cuda(non_blocking=True), msk.cuda(non_blocking=True)

    def _batches(self):
        n = len(self.images)
        starts = list(range(0, n, self.batch_size))
        if self.shuffle:
            starts = [starts[i] for i in self.rng.permutation(len(starts))]
        for s in starts:
            imgs, msks = [], []
            for i in range(s, min(s + self.batch_size, n)):
                img, msk = self._sample(i)
                imgs.append(img.float())
                msks.append(msk.float() if self.one_hot else msk)
            x, y = torch.stack(imgs), torch.stack(msks)
            if self.augment:
                params = _ops.random_augment_params(len(imgs), self.rng)
                x, y2 = _ops.augment(x, y if self.one_hot else None, params)
                y = y2 if y2 is not None else y
            yield _ops.normalize_image(x), y

    def __iter__(self):
        while True:
            yield from self._batches()
            if not self.repeat:
                return


def load_dataset(dataset_path, batch_size=1, shape=(1920, 2560), classes=3, mask_one_hot_encoded=True,
                 repeat=False, shuffle=True, augment=False, rank=None, world=None):
    """Dataset over `<dataset_path>/images` and `<dataset_path>/masks`.  With torch.distributed
    initialised each rank gets every world-th sample (the data-parallel split of the batch)."""
    root = Path(dataset_path)
    if not root.is_dir():
        raise FileNotFoundError("The directory `%s` does not exist." % root)
    for sub in ("images", "masks"):
        if not root.joinpath(sub).is_dir():
            raise FileNotFoundError("The directory `%s` does not exist." % root.joinpath(sub))
    if rank is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    return Dataset(list_files(str(root.joinpath("images"))), list_files(str(root.joinpath("masks"))), batch_size,
                   shape, classes, mask_one_hot_encoded, repeat, shuffle, augment, rank=rank, world=world)


def write_dataset(dataset, batches=None, output_path="dataset_visualization", rgb_masks=True):
    """Write the first `batches` batches of a dataset as PNG files for inspection."""
    from PIL import Image
    from utils.utils import color_classes
    out = Path(output_path)
    out.joinpath("images").mkdir(exist_ok=True, parents=True)
    out.joinpath("masks").mkdir(exist_ok=True, parents=True)
    for i, (images, masks) in enumerate(dataset):
        if batches is not None and i >= max(0, batches):
            break
        for j, (image, mask) in enumerate(zip(images.cpu().numpy(), masks.cpu().numpy())):
            Image.fromarray((image * 255).clip(0, 255).astype(np.uint8)).save(
                str(out.joinpath("images", "batch_%d_image_%d.png" % (i, j))))
            m = (mask * 127).astype(np.uint8) if mask.ndim == 3 else mask.astype(np.uint8)
            Image.fromarray(color_classes(m.copy())).save(str(out.joinpath("masks", "batch_%d_mask_%d.png" % (i, j))))
