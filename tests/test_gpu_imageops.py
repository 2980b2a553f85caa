"""Device versions of the reference's input contract and post-processing
(utils/data.py:50-80,133-138,240-295; utils/predict.py:164-166) against the numpy oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_normalize_image(pkg, dtype):
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(0)
    x = rng.integers(3, 250, (3, 37, 53, 3)).astype(dtype)
    if dtype == np.float32:
        x = x * np.float32(0.37) - np.float32(20.0)
    got = imageops.normalize_image(x).cpu().numpy()
    want = np.stack([post.normalize_image(i) for i in x])
    np.testing.assert_array_equal(got, want)           # same subtraction and division, bit for bit
    assert got.min() == 0.0 and got.max() == 1.0
    one = imageops.normalize_image(x[1]).cpu().numpy()
    np.testing.assert_array_equal(one, want[1])


def test_one_hot(pkg):
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(1)
    ids = rng.integers(0, 4, (2, 19, 23, 1)).astype(np.uint8)      # id 3 is outside 3 classes
    got = imageops.one_hot_encode(ids, 3).cpu().numpy()
    want = np.stack([post.one_hot_encode(m, 3) for m in ids])
    np.testing.assert_array_equal(got, want)
    assert got[ids[..., 0] == 3].sum() == 0


@pytest.mark.parametrize("src,dst", [((40, 56), (64, 96)), ((96, 128), (64, 64)), ((33, 47), (33, 47)),
                                     ((1200, 1600), (1920, 2560))])
def test_resize_bilinear(pkg, src, dst):
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, src + (3,)).astype(np.uint8)
    got = imageops.resize_bilinear(x, dst).cpu().numpy()
    want = post.resize_bilinear_tf(x, *dst)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-3)        # values up to 255: a few float32 ulps (FMA, scale rounding)
    gotf = imageops.resize_bilinear(x.astype(np.float32)[None], dst).cpu().numpy()[0]
    np.testing.assert_allclose(gotf, want, rtol=0, atol=1e-3)


@pytest.mark.parametrize("mode", ["tf", "cv2"])
@pytest.mark.parametrize("src,dst", [((64, 96), (40, 56)), ((64, 64), (96, 128)), ((1920, 2560), (1200, 1600))])
def test_resize_nearest(pkg, mode, src, dst):
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 2, src + (3,)) * 127).astype(np.uint8)
    got = imageops.resize_nearest(x, dst, mode=mode).cpu().numpy()
    np.testing.assert_array_equal(got, post.resize_nearest(x, *dst, mode=mode))
    g2 = imageops.resize_nearest(x[..., 0], dst, mode=mode).cpu().numpy()
    np.testing.assert_array_equal(g2, post.resize_nearest(x[..., 0], *dst, mode=mode))


def test_augment(pkg):
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(4)
    n, h, w = 6, 21, 34
    x = rng.random((n, h, w, 3), dtype=np.float32)
    m = np.eye(3, dtype=np.float32)[rng.integers(0, 3, (n, h, w))]
    cases = [dict(), dict(flip_lr=True), dict(flip_ud=True, brightness=0.15),
             dict(flip_lr=True, flip_ud=True, contrast=1.4), dict(hue=0.13, saturation=0.7, brightness=-0.1),
             dict(contrast=0.7, hue=-0.18, saturation=1.5)]
    params = np.zeros((n, 8), np.float32)
    for i, c in enumerate(cases):
        params[i] = [c.get("flip_lr", 0), c.get("flip_ud", 0), c.get("brightness", 0), c.get("contrast") or 1,
                     c.get("hue") or 0, c.get("saturation") or 1, c.get("contrast") is not None,
                     (c.get("hue") is not None) or (c.get("saturation") is not None)]
    xo, mo = imageops.augment(x, m, params)
    xo, mo = xo.cpu().numpy(), mo.cpu().numpy()
    for i, c in enumerate(cases):
        wi, wm = post.augment(x[i], m[i], **c)
        np.testing.assert_array_equal(mo[i], wm)
        np.testing.assert_allclose(xo[i], wi, rtol=0, atol=3e-5, err_msg=str(c))
    p = imageops.random_augment_params(64, np.random.default_rng(0))
    assert p.shape == (64, 8) and 10 < p[:, 0].sum() < 54 and np.all(np.abs(p[:, 2]) <= 0.2)
    assert np.all((p[:, 3] >= 0.6) & (p[:, 3] <= 1.6))


def test_predict_post_processing_chain(pkg):
    """collapse (device, fused in the head) -> nearest resize back to the original size."""
    from miseg_amd import imageops
    from oracle import post
    rng = np.random.default_rng(5)
    pr = rng.random((1, 64, 96, 3), dtype=np.float32)
    pr /= pr.sum(-1, keepdims=True)
    col = post.collapse_probabilities(pr[0])
    got = imageops.masks_to_original_size(col[None], (50, 70, 3)).cpu().numpy()[0]
    np.testing.assert_array_equal(got, post.resize_nearest(col, 50, 70, mode="cv2"))
