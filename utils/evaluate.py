"""Drop-in for the reference's `utils/evaluate.py` (/root/reference utils/evaluate.py:14-174):
metrics between mask files, and evaluation of every checkpoint of a training run on a dataset."""
import json
import re
from pathlib import Path

import numpy as np

from utils.data import list_files, load_dataset, load_image, one_hot_encode
from utils.model import METRICS, CUSTOM_OBJECTS, load_model


def evaluate_from_files(ground_truth_path, predictions_path, classes=None):
    truth = list_files(ground_truth_path, as_numpy=True)
    preds = list_files(predictions_path, as_numpy=True)
    if len(truth) != len(preds):
        raise ValueError("The number of ground truth and prediction files do not not match: '%d' != '%d'"
                         % (len(truth), len(preds)))
    infer = classes is None
    scores = {m.name: [] for m in METRICS}
    for t_path, p_path in zip(truth, preds):
        t = load_image(t_path, as_gray=True, as_numpy=True)
        p = load_image(p_path, as_gray=True, as_numpy=True)
        if infer:
            classes = np.unique(t).size
        t = one_hot_encode(t, classes=classes)[None].astype(np.float32)
        p = one_hot_encode(p, classes=classes)[None].astype(np.float32)
        for m in METRICS:
            scores[m.name].append(m(t, p).numpy())
    print("Evaluation results:")
    print("  - Number of images: %d" % len(preds))
    for m in METRICS:
        print("  - %s" % m.name)
        print("    - Mean: %s" % np.round(np.mean(scores[m.name]), 4))
        print("    - STD.: %s" % np.round(np.std(scores[m.name]), 4))
    return scores


def _epoch_of(path, model_name):
    """`<name>_e007_l...h5` -> 7; None for a file without an epoch tag."""
    tail = str(path).split(model_name)[-1]
    m = re.search(r"_e(\d{3})", tail)
    return int(m.group(1)) if m else None


def evaluate(models_paths, images_path, batch_size=1, classes=3, one_hot_encoded=True, input_shape=None,
             loss_function=CUSTOM_OBJECTS["categorical_crossentropy_plus_dice_loss"], model_name="AgNOR"):
    """Evaluate one model file or every `*.h5` of a directory whose name contains `model_name`.
    Returns (best model by f1-score, per-epoch test metrics) in the layout the reference stores in
    its train config (`test_loss`, `test_f1-score`, `test_iou_score`, indexed by epoch - 1)."""
    root = Path(models_paths)
    if root.is_file():
        paths = [str(root)]
    elif root.is_dir():
        paths = sorted(str(p) for p in root.glob("*.h5") if model_name in str(p))
    else:
        raise FileNotFoundError("No file models were found at `%s`." % models_paths)
    if not paths:
        return None, None
    # the whole test set on the calling process, also when torch.distributed is initialised (the
    # training script evaluates on its chief rank only): no per-rank split here
    dataset = load_dataset(images_path, batch_size=batch_size, shape=input_shape[:2], classes=classes,
                           mask_one_hot_encoded=one_hot_encoded, shuffle=False, rank=0, world=1)
    parent = Path(paths[0]).parent
    cfg = parent.joinpath("train_config_%s.json" % parent.name)
    if cfg.is_file():
        with open(str(cfg)) as f:
            epochs = json.load(f)["epochs"]
    else:
        epochs = max([_epoch_of(p, model_name) or 1 for p in paths])
    metrics = {"test_loss": [0] * epochs}
    for m in METRICS:
        metrics["test_%s" % m.__name__] = [0] * epochs
    best, previous = {}, None
    for path in paths:
        try:
            model = load_model(model_path=path, input_shape=input_shape, loss_function=loss_function)
            result = model.evaluate(dataset, return_dict=True)
            previous = result
        except Exception as e:                                   # unreadable checkpoint: keep going
            print("Could not open model file `%s` (%s). Replicating metrics from the previous weight." % (path, e))
            result = previous
            if result is None:
                continue
        print("Model %s" % path)
        print("  - Loss: %s" % np.round(result["loss"], 4))
        epoch = _epoch_of(path, model_name)
        slot = epoch - 1 if epoch is not None and epoch <= epochs else -1
        for key, value in result.items():
            if key != "loss":
                print("  - %s: %s" % (key, np.round(value, 4)))
            metrics["test_%s" % key][slot] = value
        if "model" not in best or result["f1-score"] > best["f1-score"]:
            best = {"model": Path(path).name, "loss": result["loss"]}
            best.update({k: v for k, v in result.items() if k != "loss"})
    if len(paths) > 1 and best:
        print("\nBest model: %s" % best["model"])
        for key, value in list(best.items())[1:]:
            print("  - %s: %s" % (key, np.round(value, 4)))
    return (best or None), metrics
