"""`patch_predict` of utils/predict.py on the HIP path: an image larger than the model's input goes
through border-anchored patches exactly as /root/reference utils/predict.py:44-70 pastes them, and
through the optional halo windows."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_halo_ranges_tile_the_axis():
    sys.path.insert(0, ROOT)
    from utils.predict import halo_ranges
    for size, patch, halo in ((1920, 512, 32), (2560, 512, 64), (512, 512, 100), (700, 512, 0), (1000, 256, 127)):
        r = halo_ranges(size, patch, halo)
        assert r[0][1] == 0 and r[-1][2] == size
        for (s, lo, hi), nxt in zip(r, r[1:] + [None]):
            assert 0 <= s and s + patch <= size and s <= lo < hi <= s + patch
            assert lo - s >= halo or s == 0                   # context before the pasted part
            assert s + patch - hi >= halo or s + patch == size
            if nxt is not None:
                assert nxt[1] == hi                           # no gap, no overlap
    assert halo_ranges(1920, 512, 0) == [(0, 0, 512), (512, 512, 1024), (1024, 1024, 1536), (1408, 1536, 1920)]
    with pytest.raises(ValueError):
        halo_ranges(100, 512, 0)
    with pytest.raises(ValueError):
        halo_ranges(1000, 512, 256)


@pytest.mark.gpu
def test_patch_predict_pastes_tiles_like_the_reference(pkg):
    sys.path.insert(0, ROOT)
    import torch
    from utils import model as um
    from utils.predict import patch_predict, halo_ranges
    m = um.make_model("resnet18", "U-Net", (64, 64, 3), 3, 1e-3, um.CUSTOM_OBJECTS[
        "categorical_crossentropy_plus_dice_loss"], um.METRICS, seed=5)
    rng = np.random.default_rng(2)
    image = rng.random((96, 160, 3), dtype=np.float32)
    got = patch_predict(m, image, (64, 64, 3)).cpu().numpy()
    assert got.shape == (96, 160, 3)
    # the reference's loop: ranges per axis, every tile forwarded alone, later tiles overwrite
    want = np.zeros_like(got)
    ys = [(0, 64), (32, 96)]
    xs = [(0, 64), (64, 128), (96, 160)]
    for y0, y1 in ys:
        for x0, x1 in xs:
            want[y0:y1, x0:x1] = m(image[None, y0:y1, x0:x1], training=False)[0].numpy()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)
    # halo windows: inner parts only
    got_h = patch_predict(m, image, (64, 64, 3), halo=8).cpu().numpy()
    want_h = np.zeros_like(got)
    for s_y, y0, y1 in halo_ranges(96, 64, 8):
        for s_x, x0, x1 in halo_ranges(160, 64, 8):
            pr = m(image[None, s_y:s_y + 64, s_x:s_x + 64], training=False)[0].numpy()
            want_h[y0:y1, x0:x1] = pr[y0 - s_y:y1 - s_y, x0 - s_x:x1 - s_x]
    np.testing.assert_array_equal(got_h, want_h)
    with pytest.raises(ValueError):
        patch_predict(m, image, (32, 32, 3))
