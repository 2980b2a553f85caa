"""Drop-in for the parts of the reference's `utils/utils.py` that the training, evaluation and
prediction scripts use (/root/reference utils/utils.py:15-130,188-252,324-358,507-538): class
colours, probability collapse, duration arithmetic, checkpoint pruning, metric plots.  The
OpenCV / labelme helpers of that module are CPU tooling outside this package.
"""
import datetime
import json
import os
from pathlib import Path

import numpy as np

_AGNOR = [[130, 130, 130], [255, 128, 0], [0, 0, 255], [128, 0, 64]]
_PAPANICOLAOU = [[130, 130, 130], [78, 121, 167], [242, 142, 43], [44, 160, 44], [200, 82, 0], [23, 190, 207],
                 [188, 189, 34], [148, 103, 189]]


def get_color_map(colormap="agnor"):
    """RGB colours per class index: "agnor" (background + 3 classes) or "papanicolaou" (+7)."""
    if colormap == "agnor":
        return [list(c) for c in _AGNOR]
    if colormap == "papanicolaou":
        return [list(c) for c in _PAPANICOLAOU]
    return None


def _extend(color_map, n):
    """More colours when a model has more classes than the map (PASCAL-VOC style bit pattern)."""
    for label in range(len(color_map), n):
        r = g = b = 0
        v = label
        for k in range(8):
            r |= ((v >> 0) & 1) << (7 - k)
            g |= ((v >> 1) & 1) << (7 - k)
            b |= ((v >> 2) & 1) << (7 - k)
            v >>= 3
        color_map.append([r, g, b])
    return color_map


def collapse_probabilities(prediction, pixel_intensity=127):
    """Per pixel, the channel that is strictly greater than all others becomes `pixel_intensity`,
    the rest 0; channels are rewritten in place one after another, as the reference does, so later
    channels are compared with the already collapsed earlier ones.  Returns uint8.  (The device
    path produces the same mask inside the network's last kernel: `Model.predict_masks`.)"""
    classes = prediction.shape[-1]
    for i in range(classes):
        wins = np.ones(prediction.shape[:2], dtype=bool)
        for j in range(classes):
            if j != i:
                wins &= prediction[:, :, i] > prediction[:, :, j]
        prediction[:, :, i] = np.where(wins, pixel_intensity, 0)
    return prediction.astype(np.uint8)


def color_classes(prediction, colormap="agnor"):
    """(H,W,C) per-class mask (non-zero = member, later classes paint over earlier ones) or (H,W)
    class indices -> (H,W,3) uint8 RGB."""
    if prediction.ndim >= 3:
        n = prediction.shape[-1]
        cmap = _extend(get_color_map("papanicolaou" if n > 4 else "agnor"), n)
        member = [prediction[:, :, i] > 0 for i in range(n)]
        for i in range(n):
            for ch in range(min(3, n)):
                prediction[:, :, ch] = np.where(member[i], cmap[i][ch], prediction[:, :, ch])
        return prediction[:, :, :3]
    values = np.unique(prediction)
    cmap = _extend(get_color_map(colormap), int(values.max()) + 1 if values.size else 0)
    out = np.zeros(prediction.shape + (3,), dtype=np.uint8)
    for v in values:
        out[prediction == v] = cmap[int(v)]
    return out


def get_duration(start, end):
    hours, rest = divmod(end - start, 3600)
    minutes, seconds = divmod(rest, 60)
    return "{:0>2}:{:0>2}:{:02.0f}".format(int(hours), int(minutes), seconds)


def add_time_delta(duration1, duration2):
    def parse(d):
        h, m, s = d.split(":")
        return datetime.timedelta(hours=int(h), minutes=int(m), seconds=int(s))
    total = (parse(duration1) + parse(duration2)).total_seconds()
    return "%d:%02d:%02d" % (total / 3600, total / 60 % 60, total % 60)


def optimize_models_storage(directory):
    """Delete every `*.h5` of a checkpoint directory except the one its train config names as best."""
    directory = Path(directory)
    cfg = directory.joinpath("train_config_%s.json" % directory.name)
    if not cfg.is_file():
        cfg = directory.joinpath("train_config.json")
        if not cfg.is_file():
            print("Train config file not found at '%s'." % directory)
            return
    with open(str(cfg)) as f:
        info = json.load(f)
    if "best_model" not in info:
        print("No best model information found in `%s`" % cfg)
        return
    best = info["best_model"]["model"]
    models = [p.name for p in directory.glob("*.h5")]
    if best not in models:
        print("The best model '%s' was not found in the directory '%s'" % (best, directory))
        return
    for name in models:
        if name != best:
            os.remove(str(directory.joinpath(name)))


def plot_metrics(train_config_path, output=None):
    """Loss / f1-score / IoU curves of a train config as PNG next to it.  Needs matplotlib; without
    it the curves stay in the JSON and a note is printed."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib is not installed: skipping the metric plots of `%s`." % train_config_path)
        return None
    with open(str(train_config_path)) as f:
        cfg = json.load(f)
    curves = dict(cfg.get("train_metrics", {}))
    curves.update(cfg.get("test_metrics", {}))
    out = Path(output) if output else Path(train_config_path).parent
    fig, axes = plt.subplots(1, 3, figsize=(18, 5))
    for ax, key in zip(axes, ("loss", "f1-score", "iou_score")):
        for prefix in ("", "val_", "test_"):
            if prefix + key in curves:
                ax.plot(range(1, len(curves[prefix + key]) + 1), curves[prefix + key], label=prefix + key)
        ax.set_xlabel("epoch")
        ax.set_title(key)
        ax.legend()
    path = out.joinpath("metrics_%s.png" % Path(train_config_path).parent.name)
    fig.savefig(str(path))
    plt.close(fig)
    return str(path)
