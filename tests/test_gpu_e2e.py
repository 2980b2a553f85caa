"""End-to-end parity of the HIP inference engine against the CPU oracle
(oracle/nets.py on the torch back-end, fp32) on seeded random-init weights with
randomised BatchNorm statistics.

Bars (BASELINE.json north_star): probabilities within a stated tolerance, strict-argmax
masks agreeing on >= 99.9 % of pixels, Dice (f1-score) and IoU against a common ground
truth within 1e-3 of the oracle's.

Two storage formats (Engine(storage=...)), two sets of bars:
  fp16 (the inference default; 11 significand bits): probabilities within 1e-2; masks agree on
    >= 99.9 % of ALL pixels at the headline size (test_headline_resnet34_unet_1920x2560, no
    rescaling of anything); the small test images hold 4-6 thousand pixels, where 0.1 % is a
    handful of near-ties, so there the bar is >= 99.9 % of the pixels whose top-2 margin exceeds the
    tolerance and >= 99.5 % overall;
  bf16 (the training format; 8 bits): probabilities within 3e-2, >= 99.9 % where the margin
    exceeds 3e-2, >= 99 % overall (99.43 % at the headline size: every one of the ~90 stored
    tensors of the network contributes about equally, see oracle/attribution.py).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("resnet18", "U-Net", (64, 96), 2, 3),
    ("resnet34", "U-Net", (96, 64), 1, 3),
    ("resnet18", "U-Net", (64, 64), 1, 1),
    ("densenet169", "Linknet", (64, 96), 1, 3),
    ("densenet169", "FPN", (64, 64), 2, 3),
    ("resnet18", "FPN", (64, 64), 4, 3),       # four images: on one, the score bars are decided by a few near-tie pixels
    ("resnet18", "Linknet", (64, 64), 1, 3),
    ("densenet121", "U-Net", (64, 64), 1, 3),
    ("resnet18", "Linknet", (96, 160), 2, 1),  # fused tail: ragged 8x64 output patches, sigmoid head
    ("resnet34", "Linknet", (64, 128), 1, 4),  # fused tail: whole patches (vector stores), four classes
    ("resnet18", "FPN", (96, 160), 1, 1),      # fused up-sampling + head: whole and ragged patches, sigmoid
    ("resnet34", "FPN", (64, 96), 2, 4),       # four classes
    ("resnet50", "U-Net", (64, 96), 1, 3),     # bottleneck units (classification_models residual_bottleneck_block)
    ("resnet50", "Linknet", (64, 64), 2, 3),
    ("vgg16", "U-Net", (64, 96), 1, 3),        # biased convolutions with ReLU, 2x2 max pools, four skips, centre blocks
    ("vgg16", "Linknet", (64, 64), 2, 3),
    ("vgg19", "FPN", (64, 64), 1, 1),
]


def make_inputs(n, h, w, seed=1234):
    from oracle import post
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (n, h, w, 3)).astype(np.uint8)
    # low-frequency structure so the network output is not pure noise
    yy, xx = np.mgrid[0:h, 0:w]
    img = (img * 0.5 + 64 * (1 + np.sin(xx / 7.0 + yy / 11.0))[None, :, :, None]).astype(np.uint8)
    return np.stack([post.normalize_image(i) for i in img])


TOL = {"fp16": (1e-2, 0.995), "bf16": (3e-2, 0.99)}     # storage -> (probability tolerance, overall agreement)


@pytest.mark.parametrize("enc,dec,hw,n,classes", CONFIGS)
def test_engine_vs_oracle(enc, dec, hw, n, classes):
    check_engine(enc, dec, hw, n, classes)


@pytest.mark.parametrize("enc,dec,hw,n,classes", [CONFIGS[0], CONFIGS[1], CONFIGS[3], CONFIGS[4]])
def test_engine_vs_oracle_bf16_storage(enc, dec, hw, n, classes):
    check_engine(enc, dec, hw, n, classes, storage="bf16")


def test_deep_bottleneck_resnet_on_bf16_storage():
    """ResNet-101 with seeded random weights (no training to keep the batch-norm scales in check) grows
    its residual stream to ~1e6 over 33 units: beyond fp16's 65504, inside bf16's range.  The engine's
    bf16 storage is the format for such a network; trained networks stay within a few tens."""
    check_engine("resnet101", "FPN", (64, 64), 1, 3, storage="bf16")


def test_model_leaves_fp16_storage_when_the_range_overflows():
    """The same network through the Model surface (`model(batch, training=False)`, /root/reference
    utils/predict.py:159): the fp16 plan yields NaN, the model rebuilds its plan on bf16 storage, warns, and
    returns the probabilities of that run."""
    import miseg_amd
    from miseg_amd import netspec
    from miseg_amd.model import Model
    from oracle import nets, ops_torch
    g = netspec.build_graph("resnet101", "FPN", (64, 64, 3), 3)
    weights = nets.init_weights(nets.build_spec("resnet101", "FPN", 3), seed=7, randomize_bn=True)
    m = Model(g, weights=weights)
    x = make_inputs(1, 64, 64)
    with pytest.warns(UserWarning, match="fp16 range"):
        got = m(x, training=False).numpy()
    assert m._storage == "bf16" and np.isfinite(got).all()
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-3)
    masks = m.predict_masks(x)
    assert masks.shape == (1, 64, 64, 3) and set(np.unique(masks)) <= {0, 127}


def headline_agreement(storage, h=1920, w=2560, want=None, x=None, weights=None, enc="resnet34", dec="U-Net"):
    """ResNet-34 U-Net, 3 classes, h x w x 3 (BASELINE config 2), seeded random-init weights with
    randomised BatchNorm statistics, nothing rescaled.  Returns (share of pixels whose mask equals
    the fp32 oracle's, |dF1|, |dIoU|, oracle probabilities) for the engine on `storage`."""
    import miseg_amd
    from miseg_amd import engine, netspec
    from oracle import nets, ops_torch, post, losses
    classes = 3
    if weights is None:
        weights = nets.init_weights(nets.build_spec(enc, dec, classes), seed=7, randomize_bn=True)
    if x is None:
        x = make_inputs(1, h, w)
    if want is None:
        with torch.no_grad():
            want = ops_torch.to_numpy(nets.forward(ops_torch, enc, dec, weights, x))
    g = netspec.build_graph(enc, dec, (h, w, 3), classes)
    eng = engine.Engine(g, weights, batch=1, storage=storage)
    got = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    m_got = eng.mask.cpu().numpy()
    del eng
    torch.cuda.empty_cache()
    m_want = np.stack([post.collapse_probabilities(p.copy()) for p in want])
    agree = float((m_got == m_want).all(axis=-1).mean())
    rng = np.random.default_rng(5)
    ids = np.repeat(np.repeat(rng.integers(0, classes, (1, h // 64, w // 64)), 64, 1), 64, 2)
    gt = np.stack([post.one_hot_encode(i[..., None], classes) for i in ids])
    df1 = abs(losses.f1_score(gt, got) - losses.f1_score(gt, want))
    diou = abs(losses.iou_score(gt, got) - losses.iou_score(gt, want))
    return agree, df1, diou, want, x, weights, float(np.abs(got - want).max())


def test_headline_resnet34_unet_1920x2560():
    """The headline configuration against the fp32 oracle, at the north-star bar as written:
    strict-argmax masks agree on >= 99.9 % of all pixels, Dice and IoU within 1e-3 (fp16 storage,
    the engine's default).  The same run on bf16 storage is recorded beside it with the bar it
    reaches (>= 99.3 %)."""
    agree, df1, diou, want, x, weights, dp = headline_agreement("fp16")
    print("fp16 storage: masks agree on %.4f %%, |dF1| %.2e, |dIoU| %.2e, max|dp| %.4f" % (100 * agree, df1, diou, dp))
    assert agree >= 0.999, agree
    assert df1 < 1e-3 and diou < 1e-3
    assert dp < 3e-2
    agree_b, df1_b, diou_b, _, _, _, dp_b = headline_agreement("bf16", want=want, x=x, weights=weights)
    print("bf16 storage: masks agree on %.4f %%, |dF1| %.2e, |dIoU| %.2e, max|dp| %.4f" % (100 * agree_b, df1_b, diou_b, dp_b))
    assert agree_b >= 0.993, agree_b
    assert df1_b < 1e-3 and diou_b < 1e-3


@pytest.mark.parametrize("enc,dec", [("densenet169", "Linknet"), ("densenet169", "FPN"), ("resnet18", "U-Net")])
def test_other_baseline_networks_1920x2560(enc, dec):
    """BASELINE's other networks at full size (DenseNet-169 LinkNet is the reference's preferred model,
    /root/reference README.md:99-107), same bar as the headline configuration: masks equal to the fp32
    oracle's on >= 99.9 % of all pixels, Dice and IoU within 1e-3, nothing rescaled.  (Measured: 99.952 %,
    100.000 %, 99.993 %; the random-weight DenseNet-169 LinkNet puts 1.8 % of its pixels within 1e-3 of a
    tie.)"""
    agree, df1, diou, _, _, _, dp = headline_agreement("fp16", enc=enc, dec=dec)
    print("%s %s: masks agree on %.4f %%, |dF1| %.2e, |dIoU| %.2e, max|dp| %.4f" % (enc, dec, 100 * agree, df1, diou, dp))
    assert agree >= 0.999, agree
    assert df1 < 1e-3 and diou < 1e-3
    assert dp < 3e-2


def test_engine_residual_stream_options():
    """The ResNet residual stream is 16-bit by default on fp16 storage and fp32 on bf16 storage
    (Engine(residual_bf16=None)); the other two combinations must clear their format's bars too."""
    check_engine("resnet34", "U-Net", (96, 64), 1, 3, residual_bf16=False)
    check_engine("resnet34", "U-Net", (96, 64), 1, 3, residual_bf16=True, storage="bf16")
    check_engine("resnet18", "Linknet", (64, 64), 1, 3, residual_bf16=True, storage="bf16")


def test_layerwise_forms_of_fused_kernels():
    """The fused kernels (U-Net / LinkNet tails, DenseNet layers) can be switched off: the layer-by-layer
    plans must clear the same bars."""
    check_engine("resnet18", "Linknet", (96, 160), 2, 3, fuse_tail=False)
    check_engine("densenet121", "Linknet", (64, 96), 1, 3, fuse_tail=False, fuse_dense=False)
    check_engine("resnet18", "U-Net", (64, 96), 2, 3, fuse_tail=False)
    check_engine("resnet18", "FPN", (96, 160), 1, 3, fuse_tail=False)
    check_engine("resnet18", "FPN", (96, 160), 2, 3, head_gather=False)      # the full-resolution fused head kernel
    check_engine("resnet18", "FPN", (64, 96), 1, 1, head_gather=False)


def check_engine(enc, dec, hw, n, classes, **engine_options):
    tol, overall = TOL[engine_options.get("storage", "fp16")]
    import miseg_amd
    from miseg_amd import engine, netspec
    from oracle import nets, ops_torch, post, losses
    h, w = hw
    g = netspec.build_graph(enc, dec, (h, w, 3), classes)
    spec = nets.build_spec(enc, dec, classes)
    assert [(l, wn, s) for l, wn, s, _ in g.weight_spec()] == \
        [(l, wn, s) for l, (_, ws) in spec.items() for wn, s, _ in ws]
    weights = nets.init_weights(spec, seed=7, randomize_bn=True)
    x = make_inputs(n, h, w)
    act = "softmax" if classes > 1 else "sigmoid"
    # random-init networks can produce logits with a spread of hundreds, which saturates the
    # softmax and turns every near-tie into a 0/1 flip; rescale the head so logits are O(1)
    head = "head_conv" if dec == "FPN" else "final_conv"
    _, logits = nets.forward(ops_torch, enc, dec, weights, x, activation=act, return_logits=True)
    spread = float(ops_torch.to_numpy(logits).std())
    if spread > 1.5:
        weights[head]["kernel"] = weights[head]["kernel"] * np.float32(1.5 / spread)
        weights[head]["bias"] = weights[head]["bias"] * np.float32(1.5 / spread)
    taps = {}
    want = ops_torch.to_numpy(nets.forward(ops_torch, enc, dec, weights, x, activation=act, taps=taps))
    eng = engine.Engine(g, weights, batch=n, **engine_options)
    got = eng.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    torch.cuda.synchronize()
    # intermediate tensors first: they localise a failure
    for name, view in eng.taps.items():
        if name in taps:
            a = view.torch().float().cpu().numpy()
            b = taps[name]
            err = np.abs(a - b).max() / (np.abs(b).max() + 1e-6)
            assert err < 0.05, "%s: relative max error %.4f" % (name, err)
    assert np.abs(got - want).max() < tol
    if classes > 1:
        m_got = eng.mask.cpu().numpy()
        m_want = np.stack([post.collapse_probabilities(p) for p in want])
        np.testing.assert_array_equal(m_got, np.stack([post.collapse_probabilities(p) for p in got]))
        agree = (m_got == m_want).all(axis=-1)
        top2 = np.sort(want, axis=-1)
        clear = (top2[..., -1] - top2[..., -2]) > tol
        assert agree[clear].mean() >= 0.999
        assert agree.mean() >= overall
        # Dice / loss against a synthetic blocky ground truth: both models must score alike
        rng = np.random.default_rng(5)
        ids = np.repeat(np.repeat(rng.integers(0, classes, (n, h // 16, w // 16)), 16, 1), 16, 2)
        gt = np.stack([post.one_hot_encode(i[..., None], classes) for i in ids])
        assert abs(losses.f1_score(gt, got) - losses.f1_score(gt, want)) < 1e-3
        assert abs(losses.iou_score(gt, got) - losses.iou_score(gt, want)) < 1e-3
        assert abs(losses.cce_dice_loss(gt, got) - losses.cce_dice_loss(gt, want)) < 2e-3


@pytest.mark.parametrize("enc,hw,n", [("resnet18", (96, 160), 2), ("densenet121", (64, 224), 1), ("resnet34", (160, 96), 1)])
@pytest.mark.parametrize("storage", ["fp16", "bf16"])
def test_linknet_fused_block_matches_layerwise(enc, hw, n, storage):
    """decoder_stage3 of a LinkNet (1x1 64 -> 16, transpose 16 -> 16, 1x1 16 -> 64, + skip; segmentation_models
    DecoderBlock, /root/reference utils/model.py:188-198) as one kernel against the same block run layer by layer:
    same operands, same rounding points, so the 64-channel outputs differ by rounding flips only.  Ragged
    8 x 64 output patches and several images are among the cases."""
    import miseg_amd
    from miseg_amd import engine, netspec
    from oracle import nets
    h, w = hw
    weights = nets.init_weights(nets.build_spec(enc, "Linknet", 3), seed=11, randomize_bn=True)
    g = netspec.build_graph(enc, "Linknet", (h, w, 3), 3)
    x = torch.from_numpy(make_inputs(n, h, w, seed=5)).cuda()
    outs = []
    for fuse in (True, False):
        eng = engine.Engine(g, weights, batch=n, storage=storage, fuse_tail=fuse)
        eng.forward(x)
        kernels = [eng.op_kernel(i) for i in range(eng.num_ops)]
        assert ("linknet_block_kernel" in kernels) == fuse
        outs.append(eng.taps["decoder_stage3_out"].torch().float().cpu().numpy())
        del eng
    a, b = outs
    assert a.shape == b.shape == (n, h // 2, w // 2, 64)
    scale = np.abs(b).max()
    ulp = 2.0 ** (-10 if storage == "fp16" else -7)
    # a rounding flip of a 16-channel intermediate moves an output by a few of its own roundings
    assert np.abs(a - b).max() <= 8 * ulp * scale, (np.abs(a - b).max(), scale)
    assert np.mean(np.abs(a - b) > ulp * np.maximum(np.abs(b), 1e-3 * scale)) < 0.02


@pytest.mark.parametrize("storage", ["fp16", "bf16"])
def test_densenet201_runs_on_both_storage_formats(storage):
    """DenseNet-201's last conv5 layers read 1824-1888 channels, more than the one-kernel dense layer takes on
    bf16 storage (1792; 3584 on fp16): those layers must fall back to their two convolutions, the others stay
    fused, and the network clears its format's bars."""
    import miseg_amd
    from miseg_amd import engine, netspec
    from oracle import nets
    check_engine("densenet201", "Linknet", (64, 64), 1, 3, storage=storage)
    g = netspec.build_graph("densenet201", "Linknet", (64, 64, 3), 3)
    eng = engine.Engine(g, nets.init_weights(nets.build_spec("densenet201", "Linknet", 3), seed=7), batch=1,
                        storage=storage)
    kernels = [eng.op_kernel(i) for i in range(eng.num_ops)]
    fused = kernels.count("dense_layer_kernel")
    assert fused == (98 if storage == "fp16" else 95), fused      # 6 + 12 + 48 + 32 layers; bf16: the last three apart


@pytest.mark.parametrize("enc,hw,steps,block", [("resnet18", (512, 640), 400, 32), ("resnet34", (1920, 2560), 300, 64)])
def test_trained_weights_agree_at_the_north_star_bar(enc, hw, steps, block):
    """A network trained by the HIP trainer (U-Net, a few hundred steps of cce+Dice on class-coloured blocky
    masks with noise; /root/reference model_train.py:284-329 is the loop) has the margins of a trained
    network rather than those of seeded random weights: the engine's masks then match the fp32 oracle's on
    >= 99.9 % of all pixels on BOTH storage formats, Dice and IoU within 1e-3 (the bar of the
    reference-matching target, which the random-weight bf16 run at full size misses).  The second case is
    the headline configuration itself: ResNet-34 U-Net, one fresh 1920x2560 image."""
    import miseg_amd
    from miseg_amd import engine, netspec, trainer, weights as W
    from oracle import nets, ops_torch, post, losses
    (h, w), n, classes = hw, 2, 3
    dev = torch.device("cuda:0")
    palette = torch.tensor([[0.15, 0.2, 0.6], [0.8, 0.3, 0.25], [0.45, 0.75, 0.4]], device=dev)

    def batch(gen, count):
        u = torch.rand((count, h // block, w // block), device=dev, generator=gen)
        ids = ((u > 0.7).long() + (u > 0.9).long()).repeat_interleave(block, 1).repeat_interleave(block, 2)
        img = palette[ids] + 0.2 * torch.randn((count, h, w, 3), device=dev, generator=gen)
        lo, hi = img.amin(dim=(1, 2, 3), keepdim=True), img.amax(dim=(1, 2, 3), keepdim=True)
        return ((img - lo) / (hi - lo)).contiguous(), torch.nn.functional.one_hot(ids, classes).float(), ids

    g = netspec.build_graph(enc, "U-Net", (h, w, 3), classes, "softmax", None)
    tr = trainer.Trainer(g, W.init_weights(g, seed=1234), n, loss="cce_dice", learning_rate=1e-3)
    gen = torch.Generator(device=dev).manual_seed(99)
    for _ in range(steps):
        x, gt, _ = batch(gen, n)
        tr.train_step(x, gt)
    wts = tr.get_weights()
    del tr
    torch.cuda.empty_cache()
    xe, gte, ids = batch(torch.Generator(device=dev).manual_seed(7), 1)
    x_np, gt_np = xe.cpu().numpy(), gte.cpu().numpy()
    with torch.no_grad():
        want = ops_torch.to_numpy(nets.forward(ops_torch, enc, "U-Net", wts, x_np))
    assert float((want.argmax(-1) == ids.cpu().numpy()).mean()) > 0.95          # it did learn the task
    m_want = np.stack([post.collapse_probabilities(p.copy()) for p in want])
    for storage in ("fp16", "bf16"):
        eng = engine.Engine(g, wts, batch=1, storage=storage)
        got = eng.forward(xe).cpu().numpy()
        agree = float((eng.mask.cpu().numpy() == m_want).all(axis=-1).mean())
        del eng
        torch.cuda.empty_cache()
        print("%s %dx%d, %s storage, trained weights: masks agree on %.4f %%" % (enc, h, w, storage, 100 * agree))
        assert agree >= 0.999, (storage, agree)
        assert abs(losses.f1_score(gt_np, got) - losses.f1_score(gt_np, want)) < 1e-3
        assert abs(losses.iou_score(gt_np, got) - losses.iou_score(gt_np, want)) < 1e-3
