#!/usr/bin/env python3
"""Evaluate model files on a dataset, or predicted mask files against ground-truth files: command
line of the reference's evaluate.py (/root/reference evaluate.py:8-97) on the MI355X path."""
import argparse
import os


def main():
    p = argparse.ArgumentParser(description="Evaluate segmentation models or predictions.")
    p.add_argument("-m", "--model", default=None, type=str, help="A model file or a directory of `*.h5` files.")
    p.add_argument("-i", "--images", default=None, type=str, help="Directory with `images` and `masks`.")
    p.add_argument("-t", "--ground-truth", default=None, type=str)
    p.add_argument("-p", "--predictions", default=None, type=str)
    p.add_argument("--input-shape", default=None, type=str, help="HEIGHTxWIDTHxCHANNELS")
    p.add_argument("--batch-size", default=1, type=int)
    p.add_argument("--classes", default=3, type=int)
    p.add_argument("--ohe", default=True, action="store_true")
    p.add_argument("-gpu", "--gpu", default="0")
    a = p.parse_args()
    if str(a.gpu).strip() == "-1":
        raise SystemExit("--gpu -1 (CPU) is not available: this implementation runs on MI355X GPUs only")
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(a.gpu))
    from utils.evaluate import evaluate, evaluate_from_files
    shape = tuple(int(v) for v in a.input_shape.lower().split("x")) if a.input_shape else None
    if a.images is not None:
        if shape is None:
            from utils.model import get_model_input_shape, load_models
            shape = get_model_input_shape(load_models(a.model, compile=False)[0])
        evaluate(models_paths=a.model, images_path=a.images, batch_size=a.batch_size, classes=a.classes,
                 one_hot_encoded=a.ohe, input_shape=shape)
    elif a.predictions is not None and a.ground_truth is not None:
        evaluate_from_files(ground_truth_path=a.ground_truth, predictions_path=a.predictions, classes=a.classes)
    else:
        p.error("give --model with --images, or --ground-truth with --predictions")


if __name__ == "__main__":
    main()
