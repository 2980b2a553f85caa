"""Generates the committed golden vectors.  They come from the torch-CPU
back-end of the oracle ("torch-CPU second opinion"); no Keras/TensorFlow run
exists for this path (the packages are not installable offline), so these
vectors pin the oracle's two back-ends and the HIP path to each other, not to
the reference's own numbers.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import nets, ops_torch, losses, post  # noqa: E402

CASES = [("resnet18", "U-Net", 64, 96, 3), ("resnet34", "U-Net", 64, 64, 3),
         ("densenet169", "Linknet", 64, 64, 3), ("densenet169", "FPN", 64, 64, 3),
         ("resnet18", "U-Net", 64, 64, 1)]
SEED_W, SEED_X = 1234, 4321


def inputs(h, w):
    rng = np.random.default_rng(SEED_X)
    img = rng.integers(0, 256, (1, h, w, 3)).astype(np.uint8)
    return np.stack([post.normalize_image(i) for i in img])


def main():
    for enc, dec, h, w, classes in CASES:
        spec = nets.build_spec(enc, dec, classes)
        wts = nets.init_weights(spec, seed=SEED_W, randomize_bn=True)
        x = inputs(h, w)
        act = "softmax" if classes > 1 else "sigmoid"
        probs, logits = nets.forward(ops_torch, enc, dec, wts, x, activation=act, return_logits=True)
        probs, logits = ops_torch.to_numpy(probs), ops_torch.to_numpy(logits)
        rng = np.random.default_rng(7)
        gt = post.one_hot_encode(rng.integers(0, max(classes, 2), (h, w, 1)), classes)[None]
        out = {"x": x, "probs": probs.astype(np.float32), "logits": logits.astype(np.float32),
               "gt": gt, "f1": losses.f1_score(gt, probs), "iou": losses.iou_score(gt, probs)}
        if classes > 1:
            out["cce_dice"] = losses.cce_dice_loss(gt, probs)
            out["mask"] = post.collapse_probabilities(probs[0])
        name = "forward_%s_%s_%dx%d_c%d.npz" % (enc, dec.replace("-", ""), h, w, classes)
        np.savez_compressed(os.path.join(HERE, name), **out)
        print(name, probs.shape, float(logits.std()))


if __name__ == "__main__":
    main()
