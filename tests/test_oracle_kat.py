"""Known-answer tests that pin the CPU oracle without any framework: parameter
counts (DenseNet-169 no-top = 12 642 880, the published Keras figure minus the
classifier), the DenseNet skip-layer indices segmentation_models uses, stage
shapes, closed-form loss values, collapse_probabilities and patch ranges from
the reference's own code, and agreement of the oracle's two op back-ends."""
import glob
import os

import numpy as np
import pytest

from oracle import losses, nets, ops_np, ops_torch, post

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("enc,dec,total", [
    ("resnet18", "U-Net", 22903084), ("resnet34", "U-Net", 33018668),
    ("densenet169", "Linknet", 17051571), ("densenet169", "FPN", 15716035)])
def test_param_counts(enc, dec, total):
    assert nets.count_params(nets.build_spec(enc, dec, 3)) == total


def test_encoder_param_counts():
    def enc_params(enc):
        spec = nets.build_spec(enc, "U-Net", 3)
        return sum(int(np.prod(s)) for n, (_, ws) in spec.items() for _, s, _ in ws
                   if not (n.startswith("decoder_") or n == "final_conv"))
    assert enc_params("densenet169") == 14307880 - (1664 * 1000 + 1000) == 12642880
    assert enc_params("resnet18") == 11186889
    assert enc_params("resnet34") == 21302473


def test_densenet_skip_indices():
    names = nets.densenet_layer_names((6, 12, 32, 32))
    assert [names[i] for i in (367, 139, 51, 4)] == ["pool4_conv", "pool3_conv", "pool2_conv", "conv1/relu"]
    assert len(names) == 595


def test_stage_shapes():
    spec = nets.build_spec("resnet34", "U-Net", 3)
    w = nets.init_weights(spec, 0)
    taps = {}
    x = np.zeros((1, 64, 96, 3), np.float32)
    out = nets.forward(ops_torch, "resnet34", "U-Net", w, x, taps=taps)
    assert tuple(out.shape) == (1, 64, 96, 3)
    assert taps["relu0"].shape == (1, 32, 48, 64)
    assert taps["stage2_unit1_relu1"].shape == (1, 16, 24, 64)
    assert taps["stage3_unit1_relu1"].shape == (1, 8, 12, 128)
    assert taps["stage4_unit1_relu1"].shape == (1, 4, 6, 256)
    assert taps["relu1"].shape == (1, 2, 3, 512)


def test_loss_closed_forms():
    n, h, w, c = 2, 6, 9, 3
    ids = (np.arange(n * h * w) % c).reshape(n, h, w)
    gt = np.stack([post.one_hot_encode(i[..., None], c) for i in ids])
    pr = np.full_like(gt, 1.0 / 3)
    assert abs(losses.categorical_crossentropy(gt, pr) - np.log(3) / 3) < 1e-6
    assert abs(losses.dice_loss(gt, pr) - 2.0 / 3) < 1e-5
    assert abs(losses.cce_dice_loss(gt, pr) - 1.032871) < 1e-5
    assert abs(losses.iou_score(gt, pr) - 0.2) < 1e-5
    assert abs(losses.f1_score(gt, gt) - 1) < 1e-9 and abs(losses.iou_score(gt, gt) - 1) < 1e-9
    assert abs(losses.categorical_crossentropy(gt, gt) - (-np.log(1 - 1e-7) / 3)) < 1e-9


def test_loss_gradient_matches_finite_differences():
    rng = np.random.default_rng(0)
    gt = post.one_hot_encode(rng.integers(0, 3, (4, 5, 1)), 3)[None]
    pr = rng.dirichlet(np.ones(3), size=(1, 4, 5))
    g = losses.cce_dice_grad(gt, pr)
    eps = 1e-6
    for idx in [(0, 0, 0, 0), (0, 1, 2, 1), (0, 3, 4, 2)]:
        p1, p2 = pr.copy(), pr.copy()
        p1[idx] += eps
        p2[idx] -= eps
        fd = (losses.cce_dice_loss(gt, p1) - losses.cce_dice_loss(gt, p2)) / (2 * eps)
        assert abs(fd - g[idx]) < 1e-5


def test_collapse_probabilities():
    p = np.array([[[0.5, 0.5, 0.0], [0.6, 0.3, 0.1], [0.2, 0.2, 0.6], [1 / 3, 1 / 3, 1 / 3]]], np.float32)
    out = post.collapse_probabilities(p, 127)
    # (0.6,0.3,0.1) -> first channel; the reference loop collapses in place, so a channel is
    # compared with the already-written 0/127 of the channels before it
    assert out[0, 1].tolist() == [127, 0, 0]
    assert out[0, 2].tolist() == [0, 0, 127]
    assert out[0, 0].tolist() == [0, 127, 0]
    assert out[0, 3].tolist() == [0, 0, 127]
    assert out.dtype == np.uint8
    rng = np.random.default_rng(1)
    q = rng.dirichlet(np.ones(3), size=(16, 16)).astype(np.float32)
    np.testing.assert_array_equal(post.collapse_probabilities(q), post.collapse_simple(q))


def test_patch_ranges():
    assert post.patch_ranges(1920, 512) == [[0, 512], [512, 1024], [1024, 1536], [1408, 1920]]
    assert post.patch_ranges(512, 512) == [[0, 512]]


def test_normalize_and_one_hot():
    img = np.array([[[10, 20, 30], [40, 50, 60]]], np.uint8)
    out = post.normalize_image(img)
    assert out.min() == 0 and out.max() == 1 and out.dtype == np.float32
    assert abs(out[0, 0, 1] - 0.2) < 1e-7
    oh = post.one_hot_encode(np.array([[[0], [2]]]), 3)
    assert oh.tolist() == [[[1, 0, 0], [0, 0, 1]]]


def test_ops_backends_agree():
    rng = np.random.default_rng(2)
    x = rng.normal(0, 1, (2, 9, 12, 8)).astype(np.float32)
    t = ops_torch.asarray(x)
    w3 = rng.normal(0, 0.2, (3, 3, 8, 5)).astype(np.float32)
    for stride, pad in [(1, 1), (2, 1), (1, 0), (2, 0)]:
        np.testing.assert_allclose(ops_np.conv2d(x, w3, stride, pad),
                                   ops_torch.to_numpy(ops_torch.conv2d(t, w3, stride, pad)), atol=1e-5)
    w7 = rng.normal(0, 0.1, (7, 7, 8, 4)).astype(np.float32)
    xe = rng.normal(0, 1, (1, 16, 20, 8)).astype(np.float32)
    np.testing.assert_allclose(ops_np.conv2d(xe, w7, 2, 3),
                               ops_torch.to_numpy(ops_torch.conv2d(ops_torch.asarray(xe), w7, 2, 3)), atol=1e-5)
    wt = rng.normal(0, 0.2, (4, 4, 6, 8)).astype(np.float32)
    np.testing.assert_allclose(ops_np.conv2d_transpose_4x4s2(x, wt),
                               ops_torch.to_numpy(ops_torch.conv2d_transpose_4x4s2(t, wt)), atol=1e-5)
    xp = np.abs(x[:, :8])
    tp = ops_torch.asarray(xp)
    np.testing.assert_array_equal(ops_np.maxpool3x3s2_pad1(xp), ops_torch.to_numpy(ops_torch.maxpool3x3s2_pad1(tp)))
    np.testing.assert_allclose(ops_np.avgpool2x2(xp), ops_torch.to_numpy(ops_torch.avgpool2x2(tp)), atol=1e-6)
    np.testing.assert_array_equal(ops_np.upsample_nearest(x, 4), ops_torch.to_numpy(ops_torch.upsample_nearest(t, 4)))
    np.testing.assert_allclose(ops_np.upsample_bilinear2x(x), ops_torch.to_numpy(ops_torch.upsample_bilinear2x(t)), atol=1e-6)
    g, b = rng.uniform(0.5, 1.5, 8).astype(np.float32), rng.normal(0, 1, 8).astype(np.float32)
    m, v = rng.normal(0, 1, 8).astype(np.float32), rng.uniform(0.5, 1.5, 8).astype(np.float32)
    np.testing.assert_allclose(ops_np.batchnorm(x, g, b, m, v, 1e-3),
                               ops_torch.to_numpy(ops_torch.batchnorm(t, g, b, m, v, 1e-3)), atol=1e-5)
    np.testing.assert_allclose(ops_np.softmax(x, 0.1), ops_torch.to_numpy(ops_torch.softmax(t, 0.1)), atol=1e-6)


def test_transpose_conv_is_adjoint_of_strided_same_conv():
    """Conv2DTranspose(4, 2, 'same') is the adjoint of a 4x4 stride-2 conv with one pixel of
    padding on each side: <conv(x), y> == <x, convT(y)>."""
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (1, 8, 10, 3)).astype(np.float64)
    y = rng.normal(0, 1, (1, 4, 5, 2)).astype(np.float64)
    k = rng.normal(0, 1, (4, 4, 3, 2)).astype(np.float32)       # HWIO of the forward conv
    fwd = ops_np.conv2d(x.astype(np.float32), k, stride=2, pad=1)
    back = ops_np.conv2d_transpose_4x4s2(y.astype(np.float32), k)   # (kh,kw,Cout=3,Cin=2)
    assert abs(float((fwd * y).sum()) - float((x * back).sum())) < 1e-3


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "forward_*.npz"))))
def test_numpy_backend_reproduces_golden(path):
    """The numpy restatement against the committed torch-CPU vectors."""
    import re
    m = re.match(r"forward_(\w+?)_(UNet|Linknet|FPN)_(\d+)x(\d+)_c(\d+)\.npz", os.path.basename(path))
    enc, dec, classes = m.group(1), {"UNet": "U-Net"}.get(m.group(2), m.group(2)), int(m.group(5))
    z = np.load(path)
    w = nets.init_weights(nets.build_spec(enc, dec, classes), seed=1234, randomize_bn=True)
    act = "softmax" if classes > 1 else "sigmoid"
    probs = ops_np.to_numpy(nets.forward(ops_np, enc, dec, w, z["x"], activation=act))
    np.testing.assert_allclose(probs, z["probs"], atol=2e-5)
    assert abs(losses.f1_score(z["gt"], probs) - float(z["f1"])) < 1e-6
    if classes > 1:
        agree = (post.collapse_probabilities(probs[0]) == z["mask"]).all(axis=-1).mean()
        assert agree > 0.9995
