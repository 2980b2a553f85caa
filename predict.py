#!/usr/bin/env python3
"""Segment images with a trained model: command line of the reference's predict.py
(/root/reference predict.py:11-223) on the MI355X path.

  python predict.py --model checkpoints/20240101/AgNOR.h5 --images slides/ --output predictions
"""
import argparse
import os
import subprocess
import sys
from pathlib import Path


def main():
    p = argparse.ArgumentParser(description="Predict segmentation masks for images.")
    p.add_argument("--model", required=True, type=str, help="Path to the model file.")
    p.add_argument("--images", required=True, type=str, help="An image file or a directory of images.")
    p.add_argument("--normalize", default=True, action="store_true")
    p.add_argument("--input-shape", default=None, type=str, help="HEIGHTxWIDTHxCHANNELS, e.g. 1920x2560x3")
    p.add_argument("--copy-images", default=False, action="store_true")
    p.add_argument("--grayscale", default=False, action="store_true")
    p.add_argument("--analyze-contours", default=False, action="store_true")
    p.add_argument("--output", default="predictions", type=str)
    p.add_argument("--analysis-output", default=None, type=str)
    p.add_argument("--record-id", default=None, type=str)
    p.add_argument("--record-class", default=None, type=str)
    p.add_argument("--bboxes", default=None, type=str)
    p.add_argument("--classify-agnor", default=False, action="store_true")
    p.add_argument("--decision-tree-model", default="agnor_classifier.joblib", type=str)
    p.add_argument("--measures-only", default=False, action="store_true")
    p.add_argument("--multi-measurements", default=False, action="store_true")
    p.add_argument("--use-bias-layer", default=False, action="store_true")
    p.add_argument("--use-bias", default=False, action="store_true")
    p.add_argument("--reclassify", default=False, action="store_true")
    p.add_argument("--remove-artifacts", default=False, action="store_true")
    p.add_argument("--gpu", default=0, help="GPU index, or several separated by commas: the sorted image list "
                                             "is then dealt over one process per GPU, nothing is exchanged.")
    a = p.parse_args()
    if a.use_bias_layer and a.use_bias:
        raise ValueError("Cannot use both bias layer and bias.")
    if str(a.gpu).strip() == "-1":
        raise SystemExit("--gpu -1 (CPU) is not available: this implementation runs on MI355X GPUs only")
    gpus = [g for g in str(a.gpu).split(",") if g.strip() != ""]
    if len(gpus) > 1 and "RANK" not in os.environ:
        # one process per GPU, started before anything here touches a GPU; the children see the same
        # arguments and find RANK / WORLD_SIZE / LOCAL_RANK in their environment
        env = dict(os.environ, HIP_VISIBLE_DEVICES=",".join(dict.fromkeys(g.strip() for g in gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(len(gpus)),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29532"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if len(gpus) == 1 and "RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", gpus[0].strip())
    from utils.data import SUPPORTED_IMAGE_TYPES
    from utils.predict import predict

    shape = tuple(int(v) for v in a.input_shape.lower().split("x")) if a.input_shape else None
    common = dict(model=a.model, normalize=a.normalize, input_shape=shape, copy_images=a.copy_images,
                  grayscale=a.grayscale, analyze_contours=a.analyze_contours, bboxes=a.bboxes,
                  classify_agnor=a.classify_agnor, decision_tree_model_path=a.decision_tree_model,
                  measures_only=a.measures_only, use_bias_layer=a.use_bias_layer, use_bias=a.use_bias,
                  reclassify=a.reclassify, remove_artifacts=a.remove_artifacts)
    if a.multi_measurements:
        folders = sorted({f.parent for f in Path(a.images).rglob("*")
                          if f.is_file() and f.suffix in SUPPORTED_IMAGE_TYPES})
        for folder in folders:
            predict(images=str(folder), record_id=folder.name, record_class=folder.parent.name,
                    output_predictions=str(Path(a.output).joinpath(folder.parent.name, folder.name)),
                    output_contour_analysis=a.output, **common)
    else:
        predict(images=a.images, output_predictions=a.output, output_contour_analysis=a.analysis_output,
                record_id=a.record_id, record_class=a.record_class, **common)


if __name__ == "__main__":
    main()
