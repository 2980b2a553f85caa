"""Model-construction API with the surface of /root/reference utils/model.py
(make_model :145-236, load_model :239-278, load_models :281-320,
get_model_input_shape :109-119, replace_model_input_shape :122-142,
CUSTOM_OBJECTS :91-100, METRICS :103-106, TemperatureScaledSoftmax :12-30),
backed by the MI355X engine in `miseg_amd` instead of tf.keras +
segmentation_models.

Differences a caller can observe:
  * encoder weights are Keras-initialised from a seed (the reference asks
    segmentation_models for `encoder_weights="imagenet"`, a download);
  * model files: Keras HDF5 (`*.h5`) is read and written by this package's own
    pure-Python HDF5 code (no h5py); other names give a single-file `.npz` container;
  * the PSPNet decoder and the PapBias layer are not provided.
"""
import os
import sys
from pathlib import Path

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import miseg_amd  # noqa: E402
from miseg_amd import losses as _losses  # noqa: E402
from miseg_amd import model as _model  # noqa: E402
from miseg_amd import netspec as _netspec  # noqa: E402
from miseg_amd.model import Adam, Model  # noqa: E402,F401


class TemperatureScaledSoftmax:
    """softmax(logits / temperature).  In this package the layer is not a
    separate pass: `make_model(..., use_tss=True)` sets the temperature of the
    softmax that the head convolution applies in its epilogue."""

    def __init__(self, temperature=1, **kwargs):
        self.temperature = temperature
        self.name = kwargs.get("name", "temperature_scaled_softmax")

    def get_config(self):
        return {"name": self.name, "temperature": self.temperature}

    def __call__(self, logits):
        import numpy as np
        z = np.asarray(logits, np.float32) / np.float32(self.temperature)
        z = z - z.max(axis=-1, keepdims=True)
        e = np.exp(z)
        return e / e.sum(axis=-1, keepdims=True)


class PapBias:
    def __init__(self, **kwargs):
        raise NotImplementedError("PapBias is OpenCV post-processing on the CPU and is not part "
                                  "of this package")


CUSTOM_OBJECTS = {
    "categorical_crossentropy": _losses.categorical_crossentropy,
    "categorical_crossentropy_plus_dice_loss": _losses.cce_dice_loss,
    "focal_loss_plus_dice_loss": _losses.categorical_focal_dice_loss,
    "focal_loss": _losses.categorical_focal_loss,
    "f1-score": _losses.f1_score,
    "iou_score": _losses.iou_score,
    "TemperatureScaledSoftmax": TemperatureScaledSoftmax,
    "PapBias": PapBias,
}

METRICS = [_losses.f1_score, _losses.iou_score]

_DEFAULT_OPTIMIZER = Adam(learning_rate=1e-5)


def get_model_input_shape(model):
    """(height, width, channels) of the model's input."""
    _, height, width, channels = model.input_shape
    return (height, width, channels)


def replace_model_input_shape(model, new_input_shape):
    """Same network and weights at a new input size (the networks are fully
    convolutional).  Follows the reference: weights out, config patched, model
    rebuilt from JSON, weights back in."""
    if not new_input_shape:
        raise TypeError("Argument `new_input_shape` is required in function `replace_model_input_shape`.")
    import json
    weights = model.get_weights()
    cfg = json.loads(model.to_json())
    cfg["config"]["layers"][0]["config"]["batch_input_shape"] = [None, *new_input_shape]
    updated = _model.model_from_json(json.dumps(cfg), custom_objects=CUSTOM_OBJECTS)
    updated.set_weights(weights)
    updated._name = model.name
    updated.device = model.device
    return updated


def make_model(encoder, decoder, input_shape, classes, learning_rate, loss_function, metrics,
               encoder_freeze=False, model_name=None, use_tss=False, seed=None):
    """Build and compile one of {resnet18/34/50/101/152, densenet121/169/201, vgg16/19} x
    {"U-Net", "Linknet", "FPN"} with the hyper-parameters the reference fixes
    (decoder filters, batch-norm use, transpose blocks, pyramid sizes)."""
    graph = _netspec.build_graph(encoder, decoder, tuple(input_shape), classes,
                                 activation="softmax" if classes > 1 else "sigmoid",
                                 temperature=0.1 if use_tss else None)
    model = Model(graph, seed=seed)
    model.encoder_freeze = bool(encoder_freeze)
    if model_name:
        model._name = model_name
    model.compile(optimizer=Adam(learning_rate=learning_rate), loss=loss_function, metrics=metrics)
    return model


def _load_h5(model_path):
    from miseg_amd import keras_h5
    return keras_h5.load_keras_h5(model_path)


def load_model(model_path, input_shape=None, loss_function=_losses.cce_dice_loss,
               optimizer=_DEFAULT_OPTIMIZER, compile=True, use_bias_layer=False):
    if not Path(model_path).is_file():
        raise FileNotFoundError(f"The model file was not found at `{model_path}`.")
    # the reference's scripts name every checkpoint *.h5 (model_train.py:187-190), so the format is
    # told from the file's first bytes: HDF5 (written by Keras) or this package's zip container
    with open(model_path, "rb") as f:
        magic = f.read(8)
    if magic == b"\x89HDF\r\n\x1a\n":
        model = _load_h5(model_path)
    else:
        model = _model.load_container(model_path)
    if input_shape:
        if tuple(input_shape) != get_model_input_shape(model):
            model = replace_model_input_shape(model, tuple(input_shape))
    if use_bias_layer:
        PapBias()
    if compile:
        model.compile(optimizer=optimizer, loss=loss_function, metrics=[METRICS])
    return model


def load_models(model_path, input_shape=None, loss_function=_losses.cce_dice_loss,
                optimizer=_DEFAULT_OPTIMIZER, compile=True):
    """Every model under a directory (or the one file given)."""
    model_path = Path(model_path)
    if model_path.is_dir():
        paths = sorted(str(p) for ext in ("*.h5", "*.npz") for p in model_path.glob(ext))
    elif model_path.is_file() and model_path.suffix in (".h5", ".npz"):
        paths = [str(model_path)]
    else:
        raise RuntimeError(f"The value of `model_path` is not a model path nor a directory "
                           f"containing one or more models: `{str(model_path)}`.")
    return [load_model(p, input_shape=input_shape, loss_function=loss_function,
                       optimizer=optimizer, compile=compile) for p in paths]
