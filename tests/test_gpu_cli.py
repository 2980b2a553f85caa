"""The command-line surface of the reference (model_train.py, predict.py, evaluate.py) end to end on
a tiny synthetic dataset: files on disk -> training run with checkpoints and train config ->
prediction PNGs -> evaluation."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_dataset(root, n_train=4, n_val=2, n_test=2, h=64, w=96):
    from PIL import Image
    rng = np.random.default_rng(0)
    for split, n in (("train", n_train), ("validation", n_val), ("test", n_test)):
        os.makedirs(os.path.join(root, split, "images"))
        os.makedirs(os.path.join(root, split, "masks"))
        for i in range(n):
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            img[:, : w // 2, 0] //= 4                      # a learnable left/right structure
            mask = np.zeros((h, w), np.uint8)
            mask[:, w // 2:] = 1
            mask[h // 4: h // 2, w // 2 + 4: w - 4] = 2
            Image.fromarray(img).save(os.path.join(root, split, "images", "s%02d.png" % i))
            Image.fromarray(mask).save(os.path.join(root, split, "masks", "s%02d_mask.png" % i))


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_train_predict_evaluate_cli(pkg, tmp_path):
    work = str(tmp_path)
    data = os.path.join(work, "data")
    _make_dataset(data)
    out = _run([os.path.join(ROOT, "model_train.py"), "--encoder", "resnet18", "--decoder", "U-Net", "--dataset", data,
                "--epochs", "2", "--batch-size", "2", "--steps", "3", "--height", "64", "--width", "96", "--lr", "1e-3",
                "--save-all", "--seed", "3", "--name", "AgNOR"], cwd=work)
    assert "Epoch 2/2" in out and "val_f1-score" in out
    runs = os.listdir(os.path.join(work, "checkpoints"))
    assert len(runs) == 1
    run_dir = os.path.join(work, "checkpoints", runs[0])
    with open(os.path.join(run_dir, "train_config_%s.json" % runs[0])) as f:
        cfg = json.load(f)
    for key in ("directory", "encoder", "decoder", "loss_function", "initial_learning_rate", "learning_rate_factor",
                "classes", "epochs", "patience", "batch_size", "steps_per_epoch", "input_shape", "augmentation",
                "one_hot_encoded", "encoder_freeze", "gpu", "metrics", "train_dataset", "validation_dataset",
                "test_dataset", "train_samples", "validation_samples", "test_samples", "save_all", "model_name",
                "description", "seed", "resume", "resume_epoch", "duration", "train_metrics", "best_model",
                "test_metrics"):
        assert key in cfg, key
    assert cfg["loss_function"] == "categorical_crossentropy_plus_dice_loss"
    assert cfg["metrics"] == ["f1-score", "iou_score"] and cfg["train_samples"] == 4
    assert set(cfg["train_metrics"]) == {"loss", "f1-score", "iou_score", "val_loss", "val_f1-score",
                                         "val_iou_score", "lr"}
    assert len(cfg["train_metrics"]["loss"]) == 2 and len(cfg["test_metrics"]["test_f1-score"]) == 2
    # optimize_models_storage kept only the best checkpoint, named as the reference names them
    kept = [f for f in os.listdir(run_dir) if f.endswith(".h5")]
    assert kept == [cfg["best_model"]["model"]] and kept[0].startswith("AgNOR_e00")
    model_path = os.path.join(run_dir, kept[0])
    # prediction: whole image at the model's size, and a larger image that goes through patches
    from PIL import Image
    big = os.path.join(work, "big")
    os.makedirs(big)
    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 256, (80, 120, 3), dtype=np.uint8)).save(os.path.join(big, "slide.png"))
    _run([os.path.join(ROOT, "predict.py"), "--model", model_path, "--images", os.path.join(data, "test", "images"),
          "--output", os.path.join(work, "pred"), "--grayscale"], cwd=work)
    preds = sorted(os.listdir(os.path.join(work, "pred")))
    assert preds == ["s00_prediction.png", "s01_prediction.png"]
    p = np.asarray(Image.open(os.path.join(work, "pred", preds[0])))
    assert p.shape == (64, 96) and p.max() <= 2
    # two processes on the one GPU of the test box (the --gpu 0,1 path: predict.py re-launches itself under
    # torch.distributed.run, every rank segments its share of the sorted files, nothing is exchanged):
    # every file is written exactly once and the masks are those of the one-process run
    _run([os.path.join(ROOT, "predict.py"), "--model", model_path, "--images", os.path.join(data, "test", "images"),
          "--output", os.path.join(work, "pred2"), "--grayscale", "--gpu", "0,0"], cwd=work)
    assert sorted(os.listdir(os.path.join(work, "pred2"))) == preds
    for name in preds:
        assert np.array_equal(np.asarray(Image.open(os.path.join(work, "pred2", name))),
                              np.asarray(Image.open(os.path.join(work, "pred", name))))
    _run([os.path.join(ROOT, "predict.py"), "--model", model_path, "--images", big, "--output",
          os.path.join(work, "pred_big")], cwd=work)
    q = np.asarray(Image.open(os.path.join(work, "pred_big", "slide_prediction.png")))
    assert q.shape == (80, 120, 3)                        # resized back to the original size, coloured
    # evaluation of the model on the test split reproduces the numbers stored at the end of training
    out = _run([os.path.join(ROOT, "evaluate.py"), "--model", model_path, "--images", os.path.join(data, "test"),
                "--input-shape", "64x96x3"], cwd=work)
    assert "f1-score" in out
    # resume for one more epoch: the config accumulates
    _run([os.path.join(ROOT, "model_train.py"), "--encoder", "resnet18", "--decoder", "U-Net", "--dataset", data,
          "--epochs", "3", "--batch-size", "2", "--steps", "2", "--height", "64", "--width", "96", "--save-all",
          "--resume", model_path, "--resume-epoch", "2"], cwd=work)
    with open(os.path.join(run_dir, "train_config_%s.json" % runs[0])) as f:
        cfg2 = json.load(f)
    assert len(cfg2["train_metrics"]["loss"]) == 3


def test_bench_two_rank_path(pkg, tmp_path):
    """bench.py under torch.distributed.run with two ranks: the line the driver's N > 1 runs print.  On the test
    box's one GPU the ranks share the device and talk over gloo (MISEG_BENCH_REHEARSAL: RCCL refuses two ranks on
    one device), small images: what is checked is the N > 1 code path -- sharded inference, the weak and the
    strong training leg with their bucketed all-reduce, the rank count the collective itself saw."""
    env = dict(os.environ, PYTHONPATH=ROOT, MISEG_BENCH_REHEARSAL="1")
    args = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--height", "256",
            "--width", "320", "--batch", "2", "--steps", "2", "--warmup", "1", "--train-batch", "4",
            "--train-global-batch", "4", "--train-steps", "1", "--train-warmup", "1", "--no-extra",
            "--no-cpu-baseline"]
    r = subprocess.run(args, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"]["world_size"] == 2 and d["world"]["ranks_in_rccl_allreduce"] == 2
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["train"]["scaling"] == "weak" and d["train"]["global_batch"] == 8 and d["train"]["value"] > 0
    assert d["train_strong"]["scaling"] == "strong" and d["train_strong"]["global_batch"] == 4
    assert "2 images per GPU" in d["train_strong"]["config"]["workload"] or "batch 2 per GPU" in d["train_strong"]["config"]["workload"]
